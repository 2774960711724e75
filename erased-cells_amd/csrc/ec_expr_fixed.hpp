// ec_expr_fixed.hpp — expression programs the library knows AHEAD OF TIME, as straight-line kernels (round 4).
//
// The interpreter (ec_expr.hpp) decodes every step once per wave and is bound by instruction issue: 0.47 of the HBM
// roofline on EVI's eight operators, 0.73 on two.  The form that is bound by memory — the program as straight-line code —
// existed only as run-time generated source compiled through hiprtc (ec_expr_jit.hip): not there for a program's first 2^31
// cell-steps, inside a stream capture that meets the program first, or on a machine without libhiprtc.  For the formulas
// every band-math caller writes, the same straight-line code is built into the library:
//
//   NDVI      (a - b) / (a + b)                                              src/gdal/rasterband.rs:148,178
//   add-mul   (a + b) * c                                                    examples/masked.rs:12 ((buf + ones) * 2.0 with c a buffer)
//   EVI       ((a - b) * k0) / (((a + b * k1) - c * k2) + k3)                the eight-operator tree of bench.py --workload evi
//   affine    a * k0 + k1                                                    scale / offset of a band
//
// each over the four cell WIDTHS (all streams of a program the same width: the bands of one raster; other mixes stay with the
// interpreter / the compiled form), 16 kernels.  As in k_fused_any and k_expr only the width is compile-time; the cell kind
// (unsigned / signed / float) of every stream, the scalars' values, the masks and the load policy are launch-uniform.
//
// A program is recognised by its TREE, not by its step list (ec_expr_fixed.hip): register names, the order in which
// independent sub-trees were scheduled and the numbering of streams and scalars do not matter — every operator is a pure
// function of its operands, rounded once, so any schedule of the same tree gives the same bits.
//
// Per step the code is what the run-time generator emits (ec_expr_jit.hip): an operator whose operands are both provably
// integers of magnitude <= 131070 (cells of a <= 16-bit stream, sums and differences of two such cells) carries no NaN test,
// and a divide of such operands is the short exact divide (ec_device.hpp div_small_int); every other step is the rounded
// f64 operator followed by cv_bin_op!'s NaN rule, tested once per pair of cells and handled out of line.
#pragma once

#include "ec_expr.hpp"

namespace ecd {

struct FStep { int op, dst, a, b; };

enum FixedId : int { kFixNdvi = 0, kFixAddMul, kFixEvi, kFixAffine, kFixCount };

template <int ID> struct FixedProg;
template <> struct FixedProg<kFixNdvi> {
    static constexpr int nsteps = 3, nstreams = 2, nscalars = 0;
    static constexpr FStep steps[3] = {{EC_SUB, 0, 0, 1}, {EC_ADD, 1, 0, 1}, {EC_DIV, 0, 4, 5}};
};
template <> struct FixedProg<kFixAddMul> {
    static constexpr int nsteps = 2, nstreams = 3, nscalars = 0;
    static constexpr FStep steps[2] = {{EC_ADD, 0, 0, 1}, {EC_MUL, 0, 4, 2}};
};
template <> struct FixedProg<kFixEvi> {
    static constexpr int nsteps = 8, nstreams = 3, nscalars = 4;
    static constexpr FStep steps[8] = {{EC_SUB, 0, 0, 1}, {EC_MUL, 0, 4, 8},    // r0 = (a - b) * k0
                                       {EC_MUL, 1, 1, 9}, {EC_ADD, 1, 0, 5},    // r1 = a + b * k1
                                       {EC_MUL, 2, 2, 10}, {EC_SUB, 1, 5, 6},   // r1 = r1 - c * k2
                                       {EC_ADD, 1, 5, 11}, {EC_DIV, 0, 4, 5}};  // r0 = r0 / (r1 + k3)
};
template <> struct FixedProg<kFixAffine> {
    static constexpr int nsteps = 2, nstreams = 1, nscalars = 2;
    static constexpr FStep steps[2] = {{EC_MUL, 0, 0, 8}, {EC_ADD, 0, 4, 9}};
};
// the trees these step lists compute, in the canonical form of ec_expr_fixed.hip (streams and scalars numbered in depth-first
// order of first use, a lone scalar operand of + or * on the right)
constexpr const char* kFixedTree[kFixCount] = {
    "(/ (- S0 S1) (+ S0 S1))",
    "(* (+ S0 S1) S2)",
    "(/ (* (- S0 S1) K0) (+ (- (+ S0 (* S1 K1)) (* S2 K2)) K3))",
    "(+ (* S0 K0) K1)",
};

// How the catalogue's streams and scalars map onto the caller's (canonical index -> the caller's index), and the caller's load
// policy bits in canonical order.
struct FixedMap {
    int8_t stream[kExprMaxStreams];
    int8_t scalar[kExprMaxScalars];
    uint8_t cacheable;
};

// is the value `ref` names, as step K reads it, provably an integer of magnitude <= 131070?  (C: the streams' cell width)
template <typename P, int C>
constexpr bool fixed_small(int K, int ref) {
    if (ref < kRefReg0) return C <= 2;
    if (ref >= kRefScalar0) return false;
    for (int j = K - 1; j >= 0; --j)
        if (P::steps[j].dst == ref - kRefReg0) {
            const FStep s = P::steps[j];
            return (s.op == EC_ADD || s.op == EC_SUB) && s.a < kRefReg0 && s.b < kRefReg0 && C <= 2;
        }
    return false;
}

template <int REF, int N>
__device__ __forceinline__ double fixed_operand(const double (&s)[kExprMaxStreams][N], const double (&r)[kExprRegs][N], const double (&sc)[kExprMaxScalars], int i) {
    if constexpr (REF < kRefReg0) return s[REF][i];
    else if constexpr (REF < kRefScalar0) return r[REF - kRefReg0][i];
    else return sc[REF - kRefScalar0];
}

template <typename P, int C, int N, int K = 0>
__device__ __forceinline__ void fixed_run(const double (&s)[kExprMaxStreams][N], const double (&sc)[kExprMaxScalars], double (&r)[kExprRegs][N]) {
    if constexpr (K < P::nsteps) {
        constexpr FStep st = P::steps[K];
        constexpr bool small = fixed_small<P, C>(K, st.a) && fixed_small<P, C>(K, st.b);
        double t[N];
        if constexpr (small && st.op == EC_DIV) {
            double a[N], b[N], y[N], e[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                a[i] = fixed_operand<st.a, N>(s, r, sc, i);
                b[i] = fixed_operand<st.b, N>(s, r, sc, i);
            }
            div_small_int_nonzero_staged<N>(a, b, t, y, e);  // the N chains of six dependent instructions overlap
#pragma unroll
            for (int i = 0; i < N; ++i) t[i] = b[i] == 0.0 ? div_by_zero(a[i]) : t[i];
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) t[i] = apply_op<st.op>(fixed_operand<st.a, N>(s, r, sc, i), fixed_operand<st.b, N>(s, r, sc, i));
            if constexpr (!small) {  // cell_op<OP, true>: the first NaN operand, quieted; the x86 default NaN when neither is one
                bool nan = false;
#pragma unroll
                for (int i = 0; i + 1 < N; i += 2) nan = nan || __builtin_isunordered(t[i], t[i + 1]);
                if constexpr (N & 1) nan = nan || t[N - 1] != t[N - 1];
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(nan) != 0, 0)) {
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const double x = fixed_operand<st.a, N>(s, r, sc, i), y = fixed_operand<st.b, N>(s, r, sc, i);
                        const uint64_t fix = (x != x) ? (f64_bits(x) | kQuietBit) : (y != y) ? (f64_bits(y) | kQuietBit) : kNegQNaN;
                        t[i] = (t[i] != t[i]) ? bits_f64(fix) : t[i];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) r[st.dst][i] = t[i];
        fixed_run<P, C, N, K + 1>(s, sc, r);
    }
}

// Tile shape of the compiled form (ec_expr_jit.hip): 2 pairs per lane, one workgroup per tile, two fronts, loads under the launch's
// policy, streaming stores; the peeled head cell / odd tail cell and the masks' AND are the interpreter's code (ec_expr.hpp), on the
// caller's own program — same cells either way.
constexpr int kFixedU = 2;

template <int ID, int C>
__global__ __launch_bounds__(kBlock) void k_expr_fixed(ExprArgs ea, FixedMap fm, double* __restrict__ out, uint8_t* __restrict__ out_mask, size_t n) {
    using P = FixedProg<ID>;
    constexpr int U = kFixedU, NC = 2 * U, NS = P::nstreams;
    using Raw = typename raw_pair<C>::type;
    const unsigned head = ea.head;
    const size_t npairs = (n - head) >> 1;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t tile = two_front_tile();
    const size_t base = tile * TILE + threadIdx.x;
    const bool full = tile * TILE + TILE <= npairs;
    D2* __restrict__ op = reinterpret_cast<D2*>(out + head);
    const Raw* b[NS];
    int kind[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        b[k] = reinterpret_cast<const Raw*>(static_cast<const char*>(ea.p[fm.stream[k]]) + size_t(head) * C);
        kind[k] = ea.dt[fm.stream[k]] >> 2;
    }
    double sc[kExprMaxScalars] = {};
#pragma unroll
    for (int k = 0; k < P::nscalars; ++k) sc[k] = ea.sc[fm.scalar[k]];

    Raw q[NS][U] = {};
    if (full) {
        policy_arms<NS>(fm.cacheable, [&](auto bits) {
            constexpr unsigned B = decltype(bits)::value;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const size_t pr = base + size_t(j) * kBlock;
                q[0][j] = load_vec<!(B & 1u)>(b[0] + pr);
                if constexpr (NS > 1) q[1][j] = load_vec<!(B & 2u)>(b[1] + pr);
                if constexpr (NS > 2) q[2][j] = load_vec<!(B & 4u)>(b[2] + pr);
            }
        });
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t pr = base + size_t(j) * kBlock;
            if (pr < npairs)
#pragma unroll
                for (int k = 0; k < NS; ++k) q[k][j] = nt_load(b[k] + pr);
        }
    }
    constexpr int last = P::steps[P::nsteps - 1].dst;
    // Long programs run chunk by chunk (the pair of cells of one store): a chunk's steps run and its store leaves as soon as the chunk's
    // loads are back, while the next chunk's may still be in flight — EVI 0.797 against 0.789 over the tile's four cells at once; the two-
    // and three-step programs lose half a point that way (fewer independent chains per step) and run over the whole tile
    // (profiles/r04/fixed_chunked_ab.md).
    if constexpr (P::nsteps > 3) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
        double s[kExprMaxStreams][2] = {}, r[kExprRegs][2];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const Raw one[1] = {q[k][j]};
            widen_pairs<C, 1>(one, kind[k], s[k]);
        }
        fixed_run<P, C, 2>(s, sc, r);
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) nt_store(D2{r[last][0], r[last][1]}, op + pr);
    }
    } else {
    double s[kExprMaxStreams][NC] = {}, r[kExprRegs][NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) widen_pairs<C, U>(q[k], kind[k], s[k]);
    fixed_run<P, C, NC>(s, sc, r);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) nt_store(D2{r[last][2 * j], r[last][2 * j + 1]}, op + pr);
    }
    }
    if (blockIdx.x == 0 && threadIdx.x < 2) {  // the peeled head cell (lane 0) and the odd tail cell (lane 1)
        const bool do_it = threadIdx.x == 0 ? head != 0 : ((n - head) & 1) != 0;
        const size_t i = threadIdx.x == 0 ? 0 : n - 1;
        if (do_it) st_cell(expr_one_cell(ea, i), out + i);
    }
    expr_mask_phase(ea, out_mask, n);
}

// Host side (ec_expr_fixed.hip): *launched = true when the program is one of the catalogue's and its kernel was launched.
ec_status expr_fixed_launch(const ExprArgs& ea, size_t n, double* out, uint8_t* out_mask, hipStream_t s, bool* launched);
int64_t expr_fixed_stat(const char* key, bool* known);
// the canonical tree of a program ("" when it has no bounded one), with the maps; `id` = its catalogue entry or -1
std::string expr_fixed_tree(const ExprArgs& ea, FixedMap* fm, int* id);

}  // namespace ecd

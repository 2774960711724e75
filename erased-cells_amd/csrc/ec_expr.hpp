// ec_expr.hpp — expression PROGRAMS over up to four operand streams, evaluated per cell in registers, in one pass.
//
// The reference evaluates an operator tree eagerly: `2.5 * (nir - red) / (nir + 6.0 * red - 7.5 * blue + 1.0)` (EVI, the
// next formula after NDVI in any band-math text) is eight passes with seven f64 temporaries.  Every intermediate of the
// eager evaluation is an f64 rounded once per operator (src/value.rs:207), so running the same operators in the same order
// on registers gives bit-identical results while no temporary touches HBM.  k_fused_any covers the fixed two-level shape
// `(x o1 y) o2 (z o3 w)`; this kernel covers any tree (any DAG) the host can schedule onto four temporaries:
//
//   streams  up to 4 buffers of any cell types (compile-time: only their byte widths, as in k_fused_any), widened once
//   scalars  up to 8 constants, widened to f64 on the host
//   program  up to 16 steps  reg[dst] = A op B,  A and B each a stream, a register (0..3) or a scalar — launch-uniform
//
// The interpreter runs per TILE, not per cell: a step decodes once per wave (scalar shifts), reads its operands out of the
// lane's register file with the VGPR index mode (uniform index: no branch), runs the op arm over the lane's 4 cells and
// files the result the same way — nothing spills.  The host allocates the registers (ec_expr.hip): a tree needs at
// most ⌈log2(leaves)⌉ + 1 live temporaries when its deeper sub-tree is evaluated first; one that needs more than four is
// refused and the caller evaluates a sub-tree eagerly first.
//
// Kernel family: k_expr<C0,C1,C2,C3>, stream load classes packed (a class 0 only after the last stream): 4 + 16 + 64 +
// 256 = 340 kernels.  Load policy, tile shape (2 pairs per lane), two fronts, nt stores: as k_fused_any.
#pragma once

#include "ec_fused_any.hpp"

namespace ecd {

constexpr int kExprMaxStreams = 4, kExprRegs = 4, kExprMaxScalars = 8, kExprMaxSteps = 16;
// operand references of a step
constexpr int kRefStream0 = 0, kRefReg0 = 4, kRefScalar0 = 8, kRefEnd = 16;
// cell PAIRS per lane per tile.  A step is decoded once per wave (≈ 45 scalar instructions: the binding cost),
// so more cells per wave amortise the decode — but the register file costs 16 VGPRs per cell and the waves of a SIMD must
// still cover each other's scalar-branch latency.  Measured on EVI at 16384² (profiles/r03/expr_kernel.md):
//   2 pairs (100 VGPRs, 5 waves/SIMD) 1.070 ms;  3 pairs (3 waves/SIMD) 0.992 ms;  4 pairs (188 VGPRs, 2 waves/SIMD) 1.139 ms.
#ifndef EC_EXPR_U
#define EC_EXPR_U 3
#endif
constexpr int kExprU = EC_EXPR_U;

struct ExprArgs {
    const void* p[kExprMaxStreams];
    const uint8_t* m[kExprMaxStreams];  // distinct masks to AND (masked form), first nmask entries
    double sc[kExprMaxScalars];
    int8_t dt[kExprMaxStreams];         // cell types of the streams
    // The program, 16 bits per step: op (bits 0-1), dst (2-3), a (4-7), b (8-11), and the host's marks — a / b IS the
    // previous step's result (12 / 13), no later step reads the register this step writes (14); step k in bits
    // 16(k&3).. of prog[k>>2].  Packed so that the kernel decodes it with scalar shifts from four SGPR pairs: per-step byte
    // arrays in the kernel argument were fetched with global_load_ubyte, a vector-memory round trip per step and wave.
    uint64_t prog[kExprMaxSteps / 4];
    int8_t nstreams, nsteps, nmask, pad_;
    uint8_t head;       // leading cells (0/1) computed singly (peel rule of the binop kernels)
    uint8_t cacheable;  // load policy: bit k = stream k, bit 4 + j = mask j (cache_plan, ec_runtime.hpp)
};

// The interpreter over N cells held by one lane (N = 4 in the tile, 1 for the peeled head / odd tail cell).
// `v[s]`: the streams' cells, already f64 (NS of them are real).
//
// The register file of cell slot i is ONE 8-lane vector `x[i]` = {stream 0..3, register 0..3}: 16 consecutive VGPRs.  A
// step's operand references are launch-uniform (SGPRs), and a uniform dynamic index into a VGPR tuple is what gfx9's VGPR
// index mode does: `s_set_gpr_idx_on sN, gpr_idx(SRC0)` + one v_mov_b32 per dword + `s_set_gpr_idx_off` — no branch, no
// scratch.  What the counters said of the forms before this one (profiles/r03/expr_kernel.md): the kernel is bound by
// instruction issue — ≈ 94 VALU and ≈ 105 SALU instructions per cell slot for the 8 steps of EVI, the CU's scalar unit 64 %
// busy and the SIMDs' VALUs 58 % — so the work per step is what counts:
//   - the previous step's result stays in fixed registers (`acc`) and the next step reads it THERE when it refers to the
//     register that step wrote (the host marks such operands, ec_expr.hip): no read-out, and no filing either when nothing
//     later reads the register;
//   - a scalar right operand is used from its SGPR pair (the op arms are instantiated per operand form: 6 forms);
//   - the NaN fix-up of cv_bin_op (ec_device.hpp cell_op) is tested once per pair of cells (`v_cmp_u_f64 r0, r1` is true
//     when either is NaN) and handled out of line;
//   - steps are decoded from a 64-bit word per four steps with two scalar instructions each.
typedef double d8 __attribute__((ext_vector_type(8)));

// bits of a packed step (ExprArgs::prog)
constexpr unsigned kStepAAcc = 1u << 12, kStepBAcc = 1u << 13, kStepNoFile = 1u << 14;

// r[i] = a(i) op b(i) for the lane's N cells; `a`, `b`: operand forms (callables of the cell slot)
template <int N, typename FA, typename FB>
__device__ __forceinline__ void expr_step_ops(unsigned op, FA&& a, FB&& b, double (&r)[N]) {
    double t[N];
    if (op == unsigned(EC_ADD)) {
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = apply_op<EC_ADD>(a(i), b(i));
    } else if (op == unsigned(EC_SUB)) {
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = apply_op<EC_SUB>(a(i), b(i));
    } else if (op == unsigned(EC_MUL)) {
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = apply_op<EC_MUL>(a(i), b(i));
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = apply_op<EC_DIV>(a(i), b(i));
    }
    bool nan = false;
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) nan = nan || __builtin_isunordered(t[i], t[i + 1]);
    if constexpr (N & 1) nan = nan || t[N - 1] != t[N - 1];
    // wave-uniform test (a lane-wise branch here would make the compiler structurize every uniform branch around it)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(nan) != 0, 0)) {  // cell_op<OP, true>: the first NaN operand, quieted; the x86 default NaN when neither is one
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double x = a(i), y = b(i);
            const uint64_t fix = (x != x) ? (f64_bits(x) | kQuietBit) : (y != y) ? (f64_bits(y) | kQuietBit) : kNegQNaN;
            t[i] = (t[i] != t[i]) ? bits_f64(fix) : t[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = t[i];
}

template <int N, int NS = 4>
__device__ __forceinline__ void expr_run(const ExprArgs& ea, const double (&v0)[N], const double (&v1)[N], const double (&v2)[N],
                                         const double (&v3)[N], double (&out)[N]) {
    d8 x[N];  // lanes 4..7 (the registers) and the lanes of absent streams start undefined: the host admits no read before a write
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[i][0] = v0[i];
        if constexpr (NS > 1) x[i][1] = v1[i];
        if constexpr (NS > 2) x[i][2] = v2[i];
        if constexpr (NS > 3) x[i][3] = v3[i];
    }
    double acc[N];  // the previous step's result
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.0;
    auto in_acc = [&](int i) { return acc[i]; };
    uint64_t cur = ea.prog[0], w1 = ea.prog[1], w2 = ea.prog[2], w3 = ea.prog[3];  // SGPR pairs
    for (int k = 0; k < ea.nsteps; ++k) {  // launch-uniform trip count
        if ((k & 3) == 0 && k != 0) {
            cur = w1;
            w1 = w2;
            w2 = w3;
        }
        const unsigned step = static_cast<unsigned>(cur) & 0xffffu;
        cur >>= 16;
        const unsigned op = step & 3u, ra = (step >> 4) & 15u, rb = (step >> 8) & 15u;
        if (step & kStepAAcc) {
            if (step & kStepBAcc) {
                expr_step_ops<N>(op, in_acc, in_acc, acc);
            } else if (rb >= unsigned(kRefScalar0)) {
                const double s = ea.sc[rb - kRefScalar0];
                expr_step_ops<N>(op, in_acc, [&](int) { return s; }, acc);
            } else {
                double B[N];
#pragma unroll
                for (int i = 0; i < N; ++i) B[i] = x[i][rb];
                expr_step_ops<N>(op, in_acc, [&](int i) { return B[i]; }, acc);
            }
        } else {
            double A[N];
            if (ra >= unsigned(kRefScalar0)) {
                const double s = ea.sc[ra - kRefScalar0];
#pragma unroll
                for (int i = 0; i < N; ++i) A[i] = s;
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) A[i] = x[i][ra];
            }
            auto in_a = [&](int i) { return A[i]; };
            if (step & kStepBAcc) {
                expr_step_ops<N>(op, in_a, in_acc, acc);
            } else if (rb >= unsigned(kRefScalar0)) {
                const double s = ea.sc[rb - kRefScalar0];
                expr_step_ops<N>(op, in_a, [&](int) { return s; }, acc);
            } else {
                double B[N];
#pragma unroll
                for (int i = 0; i < N; ++i) B[i] = x[i][rb];
                expr_step_ops<N>(op, in_a, [&](int i) { return B[i]; }, acc);
            }
        }
        if (!(step & kStepNoFile)) {
            const unsigned dst = unsigned(kRefReg0) + ((step >> 2) & 3u);
#pragma unroll
            for (int i = 0; i < N; ++i) x[i][dst] = acc[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = acc[i];  // the program's value is what its last step computed
}

// one cell of stream s as f64 (run-time typed: head / tail cells and the cell-wise kernel only)
__device__ __forceinline__ double expr_stream_cell(const ExprArgs& ea, int s, size_t i) {
    return s < ea.nstreams ? load_cell_f64(ea.p[s], ea.dt[s], i) : 0.0;
}

__device__ __forceinline__ double expr_one_cell(const ExprArgs& ea, size_t i) {
    const double v0[1] = {expr_stream_cell(ea, 0, i)}, v1[1] = {expr_stream_cell(ea, 1, i)}, v2[1] = {expr_stream_cell(ea, 2, i)},
                 v3[1] = {expr_stream_cell(ea, 3, i)};
    double o[1];
    expr_run<1>(ea, v0, v1, v2, v3, o);
    return o[0];
}

// AND of the distinct masks of the streams (src/masked/masked_buffer.rs:333 applied at every step of the eager chain)
__device__ __forceinline__ void expr_mask_phase(const ExprArgs& ea, uint8_t* __restrict__ out_mask, size_t n) {
    if (ea.nmask > 0) {
        const size_t ngroups = n / 16;
        const size_t stride = size_t(gridDim.x) * kBlock;
        u32x4* __restrict__ om = reinterpret_cast<u32x4*>(out_mask);
        for (size_t g = size_t(blockIdx.x) * kBlock + threadIdx.x; g < ngroups; g += stride) {
            u32x4 acc = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            for (int k = 0; k < ea.nmask; ++k) {
                const u32x4* mk = reinterpret_cast<const u32x4*>(ea.m[k]) + g;
                u32x4 x;
                policy_arms<1>(ea.cacheable >> (4 + k), [&](auto bits) { x = load_vec<!(decltype(bits)::value & 1u)>(mk); });
                acc &= x;
            }
            mask_store(acc, om + g);
        }
        if (blockIdx.x == 0)
            for (size_t i = ngroups * 16 + threadIdx.x; i < n; i += kBlock) {
                uint8_t acc = ld_cell(ea.m[0] + i);
                for (int k = 1; k < ea.nmask; ++k) acc &= ld_cell(ea.m[k] + i);
                st_cell(acc, out_mask + i);
            }
    }
}

template <int C0, int C1, int C2, int C3>
__global__ __launch_bounds__(kBlock) void k_expr(ExprArgs ea, double* __restrict__ out, uint8_t* __restrict__ out_mask, size_t n) {
    constexpr int U = kExprU;
    constexpr int NC = 2 * U;
    const unsigned head = ea.head;
    const size_t npairs = (n - head) >> 1;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t tile = two_front_tile();
    const size_t base = tile * TILE + threadIdx.x;
    const bool full = tile * TILE + TILE <= npairs;
    D2* __restrict__ op = reinterpret_cast<D2*>(out + head);
    const char* b0 = static_cast<const char*>(ea.p[0]) + size_t(head) * C0;
    const char* b1 = static_cast<const char*>(ea.p[1]) + size_t(head) * C1;
    const char* b2 = static_cast<const char*>(ea.p[2]) + size_t(head) * C2;
    const char* b3 = static_cast<const char*>(ea.p[3]) + size_t(head) * C3;

    typename raw_pair<C0>::type q0[U] = {};
    typename raw_pair<C1>::type q1[U] = {};
    typename raw_pair<C2>::type q2[U] = {};
    typename raw_pair<C3>::type q3[U] = {};
    if (full) {
        constexpr int kStreams = (C0 != 0) + (C1 != 0) + (C2 != 0) + (C3 != 0);  // packed: the policy bits are bits 0..kStreams-1
        policy_arms<kStreams>(ea.cacheable, [&](auto bits) {
            constexpr unsigned B = decltype(bits)::value;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const size_t pr = base + size_t(j) * kBlock;
                if constexpr (C0 != 0) q0[j] = load_vec<!(B & 1u)>(reinterpret_cast<const typename raw_pair<C0>::type*>(b0) + pr);
                if constexpr (C1 != 0) q1[j] = load_vec<!(B & 2u)>(reinterpret_cast<const typename raw_pair<C1>::type*>(b1) + pr);
                if constexpr (C2 != 0) q2[j] = load_vec<!(B & 4u)>(reinterpret_cast<const typename raw_pair<C2>::type*>(b2) + pr);
                if constexpr (C3 != 0) q3[j] = load_vec<!(B & 8u)>(reinterpret_cast<const typename raw_pair<C3>::type*>(b3) + pr);
            }
        });
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t pr = base + size_t(j) * kBlock;
            if (pr < npairs) {
                if constexpr (C0 != 0) q0[j] = nt_load(reinterpret_cast<const typename raw_pair<C0>::type*>(b0) + pr);
                if constexpr (C1 != 0) q1[j] = nt_load(reinterpret_cast<const typename raw_pair<C1>::type*>(b1) + pr);
                if constexpr (C2 != 0) q2[j] = nt_load(reinterpret_cast<const typename raw_pair<C2>::type*>(b2) + pr);
                if constexpr (C3 != 0) q3[j] = nt_load(reinterpret_cast<const typename raw_pair<C3>::type*>(b3) + pr);
            }
        }
    }
    double v0[NC] = {}, v1[NC] = {}, v2[NC] = {}, v3[NC] = {}, o[NC];
    if constexpr (C0 != 0) widen_pairs<C0, U>(q0, ea.dt[0] >> 2, v0);
    if constexpr (C1 != 0) widen_pairs<C1, U>(q1, ea.dt[1] >> 2, v1);
    if constexpr (C2 != 0) widen_pairs<C2, U>(q2, ea.dt[2] >> 2, v2);
    if constexpr (C3 != 0) widen_pairs<C3, U>(q3, ea.dt[3] >> 2, v3);
    expr_run<NC, (C0 != 0) + (C1 != 0) + (C2 != 0) + (C3 != 0)>(ea, v0, v1, v2, v3, o);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) nt_store(D2{o[2 * j], o[2 * j + 1]}, op + pr);
    }
    if (blockIdx.x == 0 && threadIdx.x < 2) {  // the peeled head cell (lane 0) and the odd tail cell (lane 1)
        const bool do_it = threadIdx.x == 0 ? head != 0 : ((n - head) & 1) != 0;
        const size_t i = threadIdx.x == 0 ? 0 : n - 1;
        if (do_it) st_cell(expr_one_cell(ea, i), out + i);
    }
    expr_mask_phase(ea, out_mask, n);
}

// Any alignment: one cell per lane (the comparison path behind ec_tune_set("unaligned_vector", 0)).
template <int UNUSED = 0>
__global__ __launch_bounds__(kBlock) void k_expr_cellwise(ExprArgs ea, double* __restrict__ out, uint8_t* __restrict__ out_mask, size_t n) {
    const size_t stride = size_t(gridDim.x) * kBlock;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        out[i] = expr_one_cell(ea, i);
        if (ea.nmask > 0) {
            uint8_t acc = ea.m[0][i];
            for (int k = 1; k < ea.nmask; ++k) acc &= ea.m[k][i];
            out_mask[i] = acc;
        }
    }
}

using ExprKernel = void (*)(ExprArgs, double*, uint8_t*, size_t);

// kernel for stream load classes (c0 fixed per translation unit; i1, i2, i3 are class INDICES 0..4; nullptr for a
// combination that is not packed)
template <int C0>
ExprKernel expr_kernel(int i1, int i2, int i3);

}  // namespace ecd

// ec_expr_fixed.hip — recognising a program of the ahead-of-time catalogue (ec_expr_fixed.hpp) and launching its kernel.
//
// Recognition is by TREE.  The caller's step list (any register names, any schedule of independent sub-trees, any numbering of
// streams and scalars — what `lazy()` of the Python mirror, `fused::tree()` of the C++ mirror or a hand-written program
// produce for one formula differs in all three) is unfolded from its last step into the expression it computes and written
// down in one canonical way:
//   * streams and scalars are renumbered in depth-first order of first use (left operand first);
//   * `k + x` and `k * x` with k a scalar and x not one are written `x + k`, `x * k` — the two differ only in which NaN wins
//     when BOTH operands are NaN (cv_bin_op! keeps the left one), so the swap is made only when k is not a NaN;
//   * nothing else is reordered: `a + b` and `b + a` stay different programs.
// Every operator is a pure function of its operands, rounded once (src/value.rs:207), so two step lists with the same tree
// produce the same bits; dead steps (results the last step does not depend on) drop out.  A register read by two later steps is
// written out twice; trees longer than the catalogue's longest are not in it and the unfolding stops.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <string>

#include "ec_expr_fixed.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

namespace ecd {

namespace {

std::atomic<int64_t> g_fixed_launches{0};
constexpr size_t kMaxTree = 96;  // the longest catalogue tree has 60 characters

struct Unfold {
    const ExprArgs& ea;
    int producer[kExprMaxSteps][2];  // for step k, operand side: the step whose result the register held when k read it (-1: not a register)
    int8_t smap[kExprMaxStreams], kmap[kExprMaxScalars];  // caller's index -> canonical
    int ns = 0, nk = 0;
    bool ok = true;
    std::string out;

    explicit Unfold(const ExprArgs& e) : ea(e) {
        std::memset(smap, -1, sizeof smap);
        std::memset(kmap, -1, sizeof kmap);
        int last_write[kExprRegs] = {-1, -1, -1, -1};
        for (int k = 0; k < ea.nsteps; ++k) {
            const unsigned st = step(k);
            const unsigned ref[2] = {(st >> 4) & 15u, (st >> 8) & 15u};
            for (int side = 0; side < 2; ++side)
                producer[k][side] = (ref[side] >= unsigned(kRefReg0) && ref[side] < unsigned(kRefScalar0)) ? last_write[ref[side] - kRefReg0] : -1;
            last_write[(st >> 2) & 3u] = k;
        }
    }
    unsigned step(int k) const { return static_cast<unsigned>(ea.prog[k >> 2] >> (16 * (k & 3))) & 0xffffu; }
    static bool is_scalar(unsigned ref) { return ref >= unsigned(kRefScalar0); }

    void leaf(unsigned ref) {
        if (ref < unsigned(kRefReg0)) {
            if (smap[ref] < 0) smap[ref] = static_cast<int8_t>(ns++);
            out += 'S';
            out += static_cast<char>('0' + smap[ref]);
        } else {
            const unsigned j = ref - kRefScalar0;
            if (kmap[j] < 0) kmap[j] = static_cast<int8_t>(nk++);
            out += 'K';
            out += static_cast<char>('0' + kmap[j]);
        }
    }
    void operand(int k, int side, unsigned ref) {
        if (producer[k][side] >= 0) node(producer[k][side]);
        else if (ref >= unsigned(kRefReg0) && ref < unsigned(kRefScalar0)) ok = false;  // a register nobody wrote (program_of refuses these)
        else leaf(ref);
    }
    void node(int k) {
        if (!ok || out.size() > kMaxTree) { ok = false; return; }
        const unsigned st = step(k), op = st & 3u;
        unsigned ref[2] = {(st >> 4) & 15u, (st >> 8) & 15u};
        int side[2] = {0, 1};
        if ((op == unsigned(EC_ADD) || op == unsigned(EC_MUL)) && is_scalar(ref[0]) && !is_scalar(ref[1])) {
            const double kv = ea.sc[ref[0] - kRefScalar0];
            if (kv == kv) {  // not a NaN: k op x == x op k bit for bit
                std::swap(ref[0], ref[1]);
                std::swap(side[0], side[1]);
            }
        }
        out += '(';
        out += "+-*/"[op];
        out += ' ';
        operand(k, side[0], ref[0]);
        out += ' ';
        operand(k, side[1], ref[1]);
        out += ')';
    }
};

using FixedKernel = void (*)(ExprArgs, FixedMap, double*, uint8_t*, size_t);

template <int ID>
FixedKernel fixed_kernel_of(int c) {
    switch (c) {
        case 1: return k_expr_fixed<ID, 1>;
        case 2: return k_expr_fixed<ID, 2>;
        case 4: return k_expr_fixed<ID, 4>;
        case 8: return k_expr_fixed<ID, 8>;
    }
    return nullptr;
}

}  // namespace

std::string expr_fixed_tree(const ExprArgs& ea, FixedMap* fm, int* id) {
    *id = -1;
    if (ea.nsteps < 1) return std::string();
    Unfold u(ea);
    u.node(ea.nsteps - 1);
    if (!u.ok || u.out.size() > kMaxTree) return std::string();
    std::memset(fm, 0, sizeof *fm);
    for (int k = 0; k < kExprMaxStreams; ++k)
        if (u.smap[k] >= 0) {
            fm->stream[u.smap[k]] = static_cast<int8_t>(k);
            fm->cacheable |= static_cast<uint8_t>(((ea.cacheable >> k) & 1u) << u.smap[k]);
        }
    for (int k = 0; k < kExprMaxScalars; ++k)
        if (u.kmap[k] >= 0) fm->scalar[u.kmap[k]] = static_cast<int8_t>(k);
    for (int i = 0; i < kFixCount; ++i)
        if (u.out == kFixedTree[i]) *id = i;
    return u.out;
}

ec_status expr_fixed_launch(const ExprArgs& ea, size_t n, double* out, uint8_t* out_mask, hipStream_t s, bool* launched) {
    *launched = false;
    if (!tuning().expr_fixed.load()) return EC_OK;
    FixedMap fm;
    int id = -1;
    (void)expr_fixed_tree(ea, &fm, &id);
    if (id < 0) return EC_OK;
    // one cell width for all the streams the tree reads (the bands of one raster), and no buffer read under two names: the catalogue's
    // S0, S1, … are distinct streams (a caller that passes one buffer twice gets the interpreter)
    const int nstreams = id == kFixNdvi ? 2 : id == kFixAffine ? 1 : 3;
    const size_t c = ecl::size_of(ea.dt[fm.stream[0]]);
    for (int k = 1; k < nstreams; ++k) {
        if (ecl::size_of(ea.dt[fm.stream[k]]) != c) return EC_OK;
        for (int j = 0; j < k; ++j)
            if (ea.p[fm.stream[j]] == ea.p[fm.stream[k]]) return EC_OK;
    }
    FixedKernel kern = nullptr;
    switch (id) {
        case kFixNdvi: kern = fixed_kernel_of<kFixNdvi>(static_cast<int>(c)); break;
        case kFixAddMul: kern = fixed_kernel_of<kFixAddMul>(static_cast<int>(c)); break;
        case kFixEvi: kern = fixed_kernel_of<kFixEvi>(static_cast<int>(c)); break;
        case kFixAffine: kern = fixed_kernel_of<kFixAffine>(static_cast<int>(c)); break;
    }
    if (!kern) return EC_OK;
    const size_t per_tile = size_t(kBlock) * kFixedU;
    const unsigned grid = grid_for((((n - ea.head) >> 1) + per_tile - 1) / per_tile);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), static_cast<unsigned>(tuning().fused_lds_kb.load()) << 10, s, ea, fm, out, out_mask, n);
    g_fixed_launches.fetch_add(1, std::memory_order_relaxed);
    *launched = true;
    return EC_OK;
}

int64_t expr_fixed_stat(const char* key, bool* known) {
    *known = !std::strcmp(key, "expr_fixed_launches");
    return *known ? g_fixed_launches.load(std::memory_order_relaxed) : 0;
}

}  // namespace ecd

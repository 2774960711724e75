// ec_fused.hip — ABI entry points of the fused two-level expression kernels (ec_fused_kernels.hpp, ec_fused_mixed.hpp):
// operand set-up (aliases, scalars, masks), dispatch by op triple; operands of two cell types run the per-slot typed
// one-pass kernel where one is instantiated, other mixes are unified (converted) first.
#include <hip/hip_runtime.h>

#include "ec_fused_mixed.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

namespace ecd {
template <int O2>
void dispatch_fused(const FusedArgs& fa, int same_dt, unsigned grid, double* out, uint8_t* out_mask, size_t n, hipStream_t s);
template <int O2>
bool dispatch_fused_mixed(const FusedArgs& fa, int pair, int pat, unsigned grid, double* out, uint8_t* om, size_t n, hipStream_t s);
}

using namespace ecd;

// Buffer operands of exactly two cell types: the one-pass kernel with per-slot typed loads (ec_fused_mixed.hpp), if
// one is instantiated for this (ordered type pair, slot pattern).  Returns true when it was launched.
static bool try_fused_mixed(FusedArgs& fa, int nops, size_t n, double* out, uint8_t* out_mask, hipStream_t s) {
    int types[2] = {-1, -1}, nt = 0;
    for (int k = 0; k < nops; ++k) {
        if (fa.is_sc[k]) continue;
        const int t = fa.dt[k];
        if (nt > 0 && t == types[0]) continue;
        if (nt > 1 && t == types[1]) continue;
        if (nt == 2) return false;  // three cell types
        types[nt++] = t;
    }
    if (nt != 2) return false;
    static const int kPairs[kFusedMixedPairs][2] = {
#define EC_ROW(IDX, AID, AT, BID, BT) {AID, BID},
        EC_FUSED_MIXED_PAIRS(EC_ROW)
#undef EC_ROW
    };
    static const int kPats4[] = {kPatABAB};
    static const int kPats3[] = {kPatAAB, kPatABA, kPatABB};
    const int* pats = nops == 4 ? kPats4 : kPats3;
    const int npats = nops == 4 ? 1 : 3;
    for (int order = 0; order < 2; ++order) {
        const int A = types[order], B = types[1 - order];
        int pair = -1;
        for (int i = 0; i < kFusedMixedPairs; ++i)
            if (kPairs[i][0] == A && kPairs[i][1] == B) pair = i;
        if (pair < 0) continue;
        for (int pi = 0; pi < npats; ++pi) {
            const int pat = pats[pi];
            bool fits = true;
            for (int k = 0; k < nops && fits; ++k)
                if (!fa.is_sc[k]) fits = (fa.dt[k] == B) == (((pat >> k) & 1) != 0);  // a scalar fits any slot
            if (!fits) continue;
            // peel one leading cell when that puts more of the 1-byte operands on even addresses (peel_head's rule)
            unsigned c0 = 0, c1 = 0;
            for (int k = 0; k < nops; ++k)
                if (!fa.is_sc[k] && fa.alias[k] == k) {
                    c0 += peel_cost(fa.p[k], ecl::size_of(fa.dt[k]), 0);
                    c1 += peel_cost(fa.p[k], ecl::size_of(fa.dt[k]), 1);
                }
            fa.head = (n >= 2 && tuning().peel && c1 < c0) ? 1 : 0;
            const size_t sa = ecl::size_of(A), sb = ecl::size_of(B);
            const size_t per_tile = size_t(kBlock) * fused_u(sa < sb ? sa : sb);
            const unsigned grid = grid_for((((n - fa.head) >> 1) + per_tile - 1) / per_tile);
            switch (fa.o2) {
                case EC_ADD: return dispatch_fused_mixed<EC_ADD>(fa, pair, pat, grid, out, out_mask, n, s);
                case EC_SUB: return dispatch_fused_mixed<EC_SUB>(fa, pair, pat, grid, out, out_mask, n, s);
                case EC_MUL: return dispatch_fused_mixed<EC_MUL>(fa, pair, pat, grid, out, out_mask, n, s);
                default: return dispatch_fused_mixed<EC_DIV>(fa, pair, pat, grid, out, out_mask, n, s);
            }
        }
    }
    return false;
}

static ec_status launch_fused(int o1, int o2, int o3, const ec_dtype dt[4], const void* const p[4],
                              const uint8_t* const masks[4], const ec_value* scalars, size_t n, double* out,
                              uint8_t* out_mask, hipStream_t s) {
    const int nops = o3 == kOpNone ? 3 : 4;
    auto op_ok = [](int o) { return o >= EC_ADD && o <= EC_DIV; };
    if (!op_ok(o1) || !op_ok(o2) || !(o3 == kOpNone || op_ok(o3))) return set_error(EC_ERR_ARG, "ec_fused: bad op");
    if (!dt || !p || !out) return set_error(EC_ERR_ARG, "ec_fused: null pointer");
    FusedArgs fa{};
    fa.o1 = static_cast<int8_t>(o1);
    fa.o2 = static_cast<int8_t>(o2);
    fa.o3 = static_cast<int8_t>(o3);
    bool aligned = aligned_to(out, 16);
    int first_buf = -1;
    for (int k = 0; k < 4; ++k) {
        const int src = k < nops ? k : 2;  // unused w mirrors z
        fa.alias[k] = static_cast<int8_t>(k);
        if (!p[src]) {  // scalar operand (impl $trt<R: Into<CellValue>>, src/buffer.rs:346-352): widened to f64 once, here
            if (!scalars || !ecl::valid(scalars[src].dtype)) return set_error(EC_ERR_ARG, "ec_fused: operand %d is neither a buffer nor a scalar", src);
            fa.is_sc[k] = 1;
            fa.sc[k] = ec_value_to_f64(&scalars[src]);
            fa.dt[k] = EC_F64;
            continue;
        }
        if (!ecl::valid(dt[src])) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_fused: bad dtype of operand %d", src);
        if (first_buf < 0) first_buf = k;
        fa.p[k] = p[src];
        fa.dt[k] = static_cast<int8_t>(dt[src]);
        for (int j = 0; j < k; ++j)
            if (!fa.is_sc[j] && fa.p[j] == fa.p[k] && fa.dt[j] == fa.dt[k]) { fa.alias[k] = static_cast<int8_t>(j); break; }
        aligned = aligned && aligned_to(fa.p[k], 16);
    }
    if (first_buf < 0) return set_error(EC_ERR_ARG, "ec_fused: at least one operand must be a buffer");
    fa.nmask = 0;
    if (masks) {
        if (!out_mask) return set_error(EC_ERR_ARG, "ec_masked_fused: null out_mask");
        aligned = aligned && aligned_to(out_mask, 16);
        for (int k = 0; k < nops; ++k) {
            if (fa.is_sc[k]) continue;  // a scalar carries no mask (masked_buffer.rs:353-364)
            if (!masks[k]) return set_error(EC_ERR_ARG, "ec_masked_fused: null mask %d", k);
            bool seen = false;
            for (int j = 0; j < fa.nmask; ++j) seen = seen || fa.m[j] == masks[k];
            if (!seen) {
                fa.m[fa.nmask++] = masks[k];
                aligned = aligned && aligned_to(masks[k], 16);
            }
        }
    }
    if (!aligned) {
        k_fused_cellwise<0><<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(fa, out, out_mask, n);
        return check_launch("fused(cellwise)");
    }
    if (tuning().fused_mixed && try_fused_mixed(fa, nops, n, out, out_mask, s)) return check_launch("fused(mixed)");
    fa.head = 0;
    // Other mixes of operand types: widen every buffer operand to the common CellType::union first (the
    // reference's `unify`, value-preserving — SURVEY App. A.1) into temporaries from the stream-ordered
    // pool, then run the same-type kernel.  Same-type calls allocate nothing.
    int u = fa.dt[first_buf];
    for (int k = 0; k < 4; ++k)
        if (!fa.is_sc[k]) u = ecl::union_of(u, fa.dt[k]);
    void* temps[4] = {nullptr, nullptr, nullptr, nullptr};
    ec_status st = EC_OK;
    for (int k = 0; k < 4 && st == EC_OK; ++k) {
        if (fa.is_sc[k] || fa.dt[k] == u) continue;
        if (fa.alias[k] != k) {  // same buffer as an earlier operand: reuse its widened copy
            fa.p[k] = fa.p[fa.alias[k]];
            fa.dt[k] = static_cast<int8_t>(u);
            continue;
        }
        st = ec_alloc_async(&temps[k], n * ecl::size_of(u), s);
        if (st != EC_OK) break;
        st = ec_convert(static_cast<ec_dtype>(fa.dt[k]), fa.p[k], static_cast<ec_dtype>(u), temps[k], n, s);
        fa.p[k] = temps[k];
        fa.dt[k] = static_cast<int8_t>(u);
    }
    if (st == EC_OK) {
        // peel one leading cell when that puts more of the (now same-typed) 1-byte operands on even addresses
        unsigned c0 = 0, c1 = 0;
        for (int k = 0; k < 4; ++k)
            if (!fa.is_sc[k] && fa.alias[k] == k) { c0 += peel_cost(fa.p[k], ecl::size_of(u), 0); c1 += peel_cost(fa.p[k], ecl::size_of(u), 1); }
        fa.head = (n >= 2 && tuning().peel && c1 < c0) ? 1 : 0;
        const size_t per_tile = size_t(kBlock) * fused_u(ecl::size_of(u));
        const unsigned grid = grid_for((((n - fa.head) >> 1) + per_tile - 1) / per_tile);
        switch (o2) {
            case EC_ADD: dispatch_fused<EC_ADD>(fa, u, grid, out, out_mask, n, s); break;
            case EC_SUB: dispatch_fused<EC_SUB>(fa, u, grid, out, out_mask, n, s); break;
            case EC_MUL: dispatch_fused<EC_MUL>(fa, u, grid, out, out_mask, n, s); break;
            default: dispatch_fused<EC_DIV>(fa, u, grid, out, out_mask, n, s); break;
        }
        st = check_launch("fused");
    }
    for (int k = 0; k < 4; ++k)
        if (temps[k]) (void)ec_free_async(temps[k], s);  // stream-ordered: released after the kernel
    return st;
}



extern "C" ec_status ec_fused(ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void* const p[4],
                              const ec_value* scalars_or_null, size_t n, double* out, ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (n == 0) return EC_OK;
    return launch_fused(o1, o2, o3, dt, p, nullptr, scalars_or_null, n, out, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" ec_status ec_masked_fused(ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void* const p[4],
                                     const uint8_t* const masks[4], const ec_value* scalars_or_null, size_t n, double* out,
                                     uint8_t* out_mask, ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (n == 0) return EC_OK;
    if (!masks) return set_error(EC_ERR_ARG, "ec_masked_fused: null masks");
    return launch_fused(o1, o2, o3, dt, p, masks, scalars_or_null, n, out, out_mask, static_cast<hipStream_t>(stream));
}

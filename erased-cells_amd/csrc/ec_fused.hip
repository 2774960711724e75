// ec_fused.hip — ABI entry points of the fused two-level expression kernels: operand set-up (aliases, scalars, masks) and
// dispatch to k_fused_any (ec_fused_any.hpp: load classes at compile time, kinds and ops launch-uniform) — one pass, no
// temporaries, for every mix of operand cell types.
#include <hip/hip_runtime.h>

#include "ec_fused_any.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

using namespace ecd;

// peel one leading cell when that puts more of the 1-byte operand streams on even addresses (peel_head's rule)
static unsigned fused_head(const FusedArgs& fa, size_t n) {
    unsigned c0 = 0, c1 = 0;
    for (int k = 0; k < 4; ++k)
        if (!fa.is_sc[k] && fa.alias[k] == k) {
            c0 += peel_cost(fa.p[k], ecl::size_of(fa.dt[k]), 0);
            c1 += peel_cost(fa.p[k], ecl::size_of(fa.dt[k]), 1);
        }
    return (n >= 2 && tuning().peel && c1 < c0) ? 1u : 0u;
}

// Any mix of operand cell types in one pass: the kernel is picked by the byte width of each slot's own stream
// (0: the slot is a scalar, an alias of an earlier slot, or the unused w of a three-operand chain).
static ec_status launch_fused_any(FusedArgs& fa, int nops, size_t n, double* out, uint8_t* out_mask, hipStream_t s) {
    int cls[4];
    bool small = fa.o2 == EC_DIV && (fa.o1 == EC_ADD || fa.o1 == EC_SUB) && (fa.o3 == EC_ADD || fa.o3 == EC_SUB || fa.o3 == kOpNone);
    size_t narrowest = 8;
    for (int k = 0; k < 4; ++k) {
        const bool own = k < nops && !fa.is_sc[k] && fa.alias[k] == k;
        const size_t bytes = own ? ecl::size_of(fa.dt[k]) : 0;
        cls[k] = fused_class_index(bytes);
        if (own && bytes < narrowest) narrowest = bytes;
        if (k < nops) small = small && !fa.is_sc[k] && ecl::is_integral(fa.dt[k]) && ecl::size_of(fa.dt[k]) <= 2;
    }
    fa.small = small ? 1 : 0;
    fa.head = static_cast<uint8_t>(fused_head(fa, n));
    size_t stream_bytes[8];
    for (int k = 0; k < 4; ++k) stream_bytes[k] = n * size_t(fused_class_bytes(cls[k]));  // 0: no stream of its own
    for (int j = 0; j < 4; ++j) stream_bytes[4 + j] = j < fa.nmask ? n : 0;
    fa.cacheable = static_cast<uint8_t>(cache_plan(stream_bytes, 8, n * sizeof(double)));
    const size_t per_tile = size_t(kBlock) * fused_u(narrowest);
    const unsigned grid = grid_for((((n - fa.head) >> 1) + per_tile - 1) / per_tile);
    FusedAnyKernel kern = nullptr;
    switch (cls[0]) {
        case 0: kern = fused_any_kernel<0>(cls[1], cls[2], cls[3]); break;
        case 1: kern = fused_any_kernel<1>(cls[1], cls[2], cls[3]); break;
        case 2: kern = fused_any_kernel<2>(cls[1], cls[2], cls[3]); break;
        case 3: kern = fused_any_kernel<4>(cls[1], cls[2], cls[3]); break;
        default: kern = fused_any_kernel<8>(cls[1], cls[2], cls[3]); break;
    }
    if (!kern) return set_error(EC_ERR_ARG, "ec_fused: no kernel for load classes %d %d %d %d", cls[0], cls[1], cls[2], cls[3]);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), static_cast<unsigned>(tuning().fused_lds_kb.load()) << 10, s, fa, out, out_mask, n);
    return check_launch("fused(any)");
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
    if (tuning().fused_mixed != 0) return launch_fused_any(fa, nops, n, out, out_mask, s);
    // fused_mixed == 0 — the comparison path of rounds 1-2, kept for A/B runs and as a second implementation the tests
    // hold the one-pass form against: widen every buffer operand to the common CellType::union first (the reference's
    // `unify`, value-preserving — SURVEY App. A.1) into temporaries from the stream-ordered pool, then run the kernel on
    // operands of one cell type.  Same-type calls convert and allocate nothing.
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
    if (st == EC_OK) st = launch_fused_any(fa, nops, n, out, out_mask, s);
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

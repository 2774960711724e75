// ec_fused.hip — ABI entry points of the fused two-level expression kernel (ec_fused_kernels.hpp).
#include <hip/hip_runtime.h>

#include "ec_fused_kernels.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

namespace ecd {
template <int O2>
void dispatch_fused(const FusedArgs& fa, int same_dt, unsigned grid, double* out, uint8_t* out_mask, size_t n, hipStream_t s);
}

using namespace ecd;

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
    bool aligned = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
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
        aligned = aligned && (reinterpret_cast<uintptr_t>(fa.p[k]) & 15u) == 0;
    }
    if (first_buf < 0) return set_error(EC_ERR_ARG, "ec_fused: at least one operand must be a buffer");
    fa.nmask = 0;
    if (masks) {
        if (!out_mask) return set_error(EC_ERR_ARG, "ec_masked_fused: null out_mask");
        aligned = aligned && (reinterpret_cast<uintptr_t>(out_mask) & 15u) == 0;
        for (int k = 0; k < nops; ++k) {
            if (fa.is_sc[k]) continue;  // a scalar carries no mask (masked_buffer.rs:353-364)
            if (!masks[k]) return set_error(EC_ERR_ARG, "ec_masked_fused: null mask %d", k);
            bool seen = false;
            for (int j = 0; j < fa.nmask; ++j) seen = seen || fa.m[j] == masks[k];
            if (!seen) {
                fa.m[fa.nmask++] = masks[k];
                aligned = aligned && (reinterpret_cast<uintptr_t>(masks[k]) & 15u) == 0;
            }
        }
    }
    if (aligned) {
        const size_t tiles = ((n >> 1) + size_t(kBlock) * kFusedU - 1) / (size_t(kBlock) * kFusedU);
        bool same = true;  // all buffer operands of one cell type?
        for (int k = 0; k < 4; ++k) same = same && (fa.is_sc[k] || fa.dt[k] == fa.dt[first_buf]);
        const int same_dt = same ? fa.dt[first_buf] : -1;
        const unsigned grid = grid_for(tiles);
        switch (o2) {
            case EC_ADD: dispatch_fused<EC_ADD>(fa, same_dt, grid, out, out_mask, n, s); break;
            case EC_SUB: dispatch_fused<EC_SUB>(fa, same_dt, grid, out, out_mask, n, s); break;
            case EC_MUL: dispatch_fused<EC_MUL>(fa, same_dt, grid, out, out_mask, n, s); break;
            default: dispatch_fused<EC_DIV>(fa, same_dt, grid, out, out_mask, n, s); break;
        }
    } else {
        k_fused_cellwise<0><<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(fa, out, out_mask, n);
    }
    return check_launch("fused");
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

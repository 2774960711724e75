// ec_fused.hip — ABI entry points of the fused two-level expression kernel (ec_fused_kernels.hpp).
#include <hip/hip_runtime.h>

#include "ec_fused_kernels.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

using namespace ecd;

static ec_status launch_fused(int o1, int o2, int o3, const ec_dtype dt[4], const void* const p[4],
                              const uint8_t* const masks[4], size_t n, double* out, uint8_t* out_mask, hipStream_t s) {
    const int nops = o3 == kOpNone ? 3 : 4;
    auto op_ok = [](int o) { return o >= EC_ADD && o <= EC_DIV; };
    if (!op_ok(o1) || !op_ok(o2) || !(o3 == kOpNone || op_ok(o3))) return set_error(EC_ERR_ARG, "ec_fused: bad op");
    if (!dt || !p || !out) return set_error(EC_ERR_ARG, "ec_fused: null pointer");
    FusedArgs fa{};
    fa.o1 = static_cast<int8_t>(o1);
    fa.o2 = static_cast<int8_t>(o2);
    fa.o3 = static_cast<int8_t>(o3);
    bool aligned = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    for (int k = 0; k < 4; ++k) {
        const int src = k < nops ? k : 2;  // unused w mirrors z
        if (!ecl::valid(dt[src])) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_fused: bad dtype of operand %d", src);
        if (!p[src]) return set_error(EC_ERR_ARG, "ec_fused: null operand %d", src);
        fa.p[k] = p[src];
        fa.dt[k] = static_cast<int8_t>(dt[src]);
        fa.alias[k] = static_cast<int8_t>(k);
        for (int j = 0; j < k; ++j)
            if (fa.p[j] == fa.p[k] && fa.dt[j] == fa.dt[k]) { fa.alias[k] = static_cast<int8_t>(j); break; }
        aligned = aligned && (reinterpret_cast<uintptr_t>(fa.p[k]) & 15u) == 0;
    }
    fa.nmask = 0;
    if (masks) {
        if (!out_mask) return set_error(EC_ERR_ARG, "ec_masked_fused: null out_mask");
        aligned = aligned && (reinterpret_cast<uintptr_t>(out_mask) & 15u) == 0;
        for (int k = 0; k < nops; ++k) {
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
        const bool same = fa.dt[0] == fa.dt[1] && fa.dt[0] == fa.dt[2] && fa.dt[0] == fa.dt[3];
        if (same) {
            switch (fa.dt[0]) {
#define EC_ROW(ID, T) case ID: k_fused_same<T><<<grid_for(tiles), kBlock, 0, s>>>(fa, out, out_mask, n); break;
                EC_WITH_CT(EC_ROW)
#undef EC_ROW
            }
        } else {
            k_fused<<<grid_for(tiles), kBlock, 0, s>>>(fa, out, out_mask, n);
        }
    } else {
        k_fused_cellwise<<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(fa, out, out_mask, n);
    }
    return check_launch("fused");
}

extern "C" ec_status ec_fused(ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void* const p[4], size_t n,
                              double* out, ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (n == 0) return EC_OK;
    return launch_fused(o1, o2, o3, dt, p, nullptr, n, out, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" ec_status ec_masked_fused(ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void* const p[4],
                                     const uint8_t* const masks[4], size_t n, double* out, uint8_t* out_mask,
                                     ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (n == 0) return EC_OK;
    if (!masks) return set_error(EC_ERR_ARG, "ec_masked_fused: null masks");
    return launch_fused(o1, o2, o3, dt, p, masks, n, out, out_mask, static_cast<hipStream_t>(stream));
}

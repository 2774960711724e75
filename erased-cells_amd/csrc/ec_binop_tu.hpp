// ec_binop_tu.hpp — launchers for one arithmetic op over all 100 operand-type
// pairs (the `with_ct!` × `with_ct!` product the reference test
// src/buffer.rs:595-614 walks).  Each of ec_binop_{add,sub,mul,div}.hip includes
// this with EC_TU_OP set, so the four ops compile in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "ec_binop_kernels.hpp"
#include "ec_runtime.hpp"

namespace ecd {

// dynamic LDS bytes of a launch: the knob's value when it is set, else the family's rule (KiB)
static inline unsigned lds_cap(int knob, int rule_kb) { return static_cast<unsigned>(knob >= 0 ? knob : rule_kb) << 10; }

template <typename L, typename R, int OP>
static ec_status launch_binop_pair(const void* l, const void* r, size_t n, double* out, hipStream_t s) {
    const Tuning& tu = tuning();
    const L* lp = static_cast<const L*>(l);
    const R* rp = static_cast<const R*>(r);
    constexpr int U = kBinopU;
    if (!aligned16(l, r, out)) {
        k_binop_cellwise<L, R, OP><<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(lp, rp, out, n);
        return check_launch("binop(cellwise)");
    }
    constexpr bool kCanStage = Staged<L>::value || Staged<R>::value;
    const int variant = tu.binop_variant.load();
    const size_t stream_bytes[2] = {n * sizeof(L), l == r ? 0 : n * sizeof(R)};  // l == r: one stream, read twice
    unsigned policy = cache_plan(stream_bytes, 2, n * sizeof(double));
    if (l == r && (policy & 1u)) policy |= 2u;
    // The LDS-staged variant by RULE (binop_variant = -1, the default): an 8-byte operand against one of <= 4 bytes — the operators of an
    // eager chain once its first result exists (u16 + f64, f64 * f32) — when no operand is kept cacheable (the staged kernel loads nt).
    // The narrow operand arrives in one 16-byte load per lane instead of two of 2-8 bytes, and with that the kernel tolerates — wants —
    // five resident workgroups per CU instead of eight, which is what its 8 B/cell store stream likes (profiles/r04/store_stream.md):
    // f64 . u16 and u16 . f64 0.80 -> 0.82, f64 . f32 0.785 -> 0.80 with every byte from HBM (the sweeps, which peak 1-2 points higher, and
    // the final A/B: profiles/r04/lds_variant_rule.md).  Every other pair measured at or below the direct kernel and stays there.
    constexpr bool kRulePair = kCanStage && ((sizeof(L) == 8) != (sizeof(R) == 8)) && (sizeof(L) <= 4 || sizeof(R) <= 4);
    const bool by_rule = variant < 0 && kRulePair && policy == 0 && n >= (size_t(1) << 20);
    if (kCanStage && (variant == 1 || by_rule)) {
        const size_t tiles = (n / kLdsWaveCells + kWavesPerBlock - 1) / kWavesPerBlock;
        // five workgroups per CU: 160 KiB of LDS in granules of 1280 bytes -> at most 32000 bytes per workgroup, slabs included
        constexpr unsigned kStatic = kWavesPerBlock * unsigned(Staged<L>::kSlabBytes + Staged<R>::kSlabBytes);
        const int knob = tu.binop_lds_kb.load();
        const unsigned lds = knob >= 0 ? unsigned(knob) << 10 : by_rule ? 32000u - kStatic : 0u;
        k_binop_lds<L, R, OP, kNtStore, kNtLoad><<<grid_for(tiles), kBlock, lds, s>>>(lp, rp, out, n);
        if (by_rule) lds_rule_launches().fetch_add(1, std::memory_order_relaxed);
        return check_launch(by_rule ? "binop(lds, by rule)" : "binop(lds)");
    }
    const unsigned head = peel_head(l, sizeof(L), r, sizeof(R), n);
    const size_t tiles = (((n - head) >> 1) + size_t(kBlock) * U - 1) / (size_t(kBlock) * U);
    k_binop_direct<L, R, OP, U, kNtStore, kNtLoad><<<grid_for(tiles), kBlock, lds_cap(tuning().binop_lds_kb.load(), sizeof(L) == 8 && sizeof(R) == 8 ? 48 : 0), s>>>(lp, rp, out, n, head | (policy << 8));
    return check_launch("binop(direct)");
}

template <typename L, typename R, int OP>
static ec_status launch_masked_pair(const void* l, const uint8_t* lm, const void* r, const uint8_t* rm, size_t n,
                                    double* out, uint8_t* om, hipStream_t s) {
    const Tuning& tu = tuning();
    const L* lp = static_cast<const L*>(l);
    const R* rp = static_cast<const R*>(r);
    constexpr int U = kBinopU;
    if (!aligned16(l, r, out) || !aligned16(lm, rm, om)) {
        k_masked_binop_cellwise<L, R, OP><<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(lp, lm, rp, rm, out, om, n);
        return check_launch("masked_binop(cellwise)");
    }
    constexpr bool kCanStage = Staged<L>::value || Staged<R>::value;
    if (kCanStage && tu.binop_variant == 1) {
        const size_t tiles = (n / kLdsWaveCells + kWavesPerBlock - 1) / kWavesPerBlock;
        k_masked_binop<L, R, OP, U, kNtStore, kNtLoad, true><<<grid_for(tiles), kBlock, lds_cap(tu.binop_lds_kb.load(), 0), s>>>(lp, lm, rp, rm, out, om, n, 0u);
        return check_launch("masked_binop(lds)");
    }
    const unsigned head = peel_head(l, sizeof(L), r, sizeof(R), n);
    const size_t tiles = (((n - head) >> 1) + size_t(kBlock) * U - 1) / (size_t(kBlock) * U);
    const size_t stream_bytes[4] = {n * sizeof(L), l == r ? 0 : n * sizeof(R), n, lm == rm ? 0 : n};
    unsigned policy = cache_plan(stream_bytes, 4, n * sizeof(double));
    if (l == r && (policy & 1u)) policy |= 2u;
    if (lm == rm && (policy & 4u)) policy |= 8u;
    k_masked_binop<L, R, OP, U, kNtStore, kNtLoad, false><<<grid_for(tiles), kBlock, 0, s>>>(lp, lm, rp, rm, out, om, n, head | (policy << 8));
    return check_launch("masked_binop(direct)");
}


template <typename L, int OP>
static ec_status launch_scalar(const void* l, double rhs, size_t n, double* out, hipStream_t s) {
    const L* lp = static_cast<const L*>(l);
    constexpr int U = kBinopU;
    if (!aligned16(l, out, out)) {
        k_binop_scalar_cellwise<L, OP><<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(lp, rhs, out, n);
        return check_launch("binop_scalar(cellwise)");
    }
    const unsigned head = peel_head(l, sizeof(L), nullptr, 0, n);
    const size_t tiles = (((n - head) >> 1) + size_t(kBlock) * U - 1) / (size_t(kBlock) * U);
    const size_t stream_bytes[1] = {n * sizeof(L)};
    const unsigned policy = cache_plan(stream_bytes, 1, n * sizeof(double));
    const unsigned lds = lds_cap(tuning().scalar_lds_kb.load(), sizeof(L) == 8 ? 32 : 0);
    if constexpr (!is_fp<L>::value) {
        // integer cells and a finite scalar (non-zero for a divide): no result can be a NaN — the form without the NaN rule
        if (std::isfinite(rhs) && !(OP == EC_DIV && rhs == 0.0)) {
            k_binop_scalar_direct<L, OP, U, kNtStore, kNtLoad, false><<<grid_for(tiles), kBlock, lds, s>>>(lp, rhs, out, n, head | (policy << 8));
            return check_launch("binop_scalar(direct, no NaN possible)");
        }
    }
    k_binop_scalar_direct<L, OP, U, kNtStore, kNtLoad><<<grid_for(tiles), kBlock, lds, s>>>(lp, rhs, out, n, head | (policy << 8));
    return check_launch("binop_scalar(direct)");
}

// Type-erased dispatch: the 10-way × 10-way match `with_ct!` stamps in the reference.
template <int OP>
ec_status dispatch_binop(int lt, const void* l, int rt, const void* r, size_t n, double* out, hipStream_t s) {
#define EC_ROW(LID, LT)                                                                         \
    case LID:                                                                                   \
        switch (rt) {                                                                           \
            case EC_U8: return launch_binop_pair<LT, uint8_t, OP>(l, r, n, out, s);             \
            case EC_U16: return launch_binop_pair<LT, uint16_t, OP>(l, r, n, out, s);           \
            case EC_U32: return launch_binop_pair<LT, uint32_t, OP>(l, r, n, out, s);           \
            case EC_U64: return launch_binop_pair<LT, uint64_t, OP>(l, r, n, out, s);           \
            case EC_I8: return launch_binop_pair<LT, int8_t, OP>(l, r, n, out, s);              \
            case EC_I16: return launch_binop_pair<LT, int16_t, OP>(l, r, n, out, s);            \
            case EC_I32: return launch_binop_pair<LT, int32_t, OP>(l, r, n, out, s);            \
            case EC_I64: return launch_binop_pair<LT, int64_t, OP>(l, r, n, out, s);            \
            case EC_F32: return launch_binop_pair<LT, float, OP>(l, r, n, out, s);              \
            case EC_F64: return launch_binop_pair<LT, double, OP>(l, r, n, out, s);             \
        }                                                                                       \
        break;
    switch (lt) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "binop: bad dtype");
}

template <int OP>
ec_status dispatch_masked_binop(int lt, const void* l, const uint8_t* lm, int rt, const void* r, const uint8_t* rm,
                                size_t n, double* out, uint8_t* om, hipStream_t s) {
#define EC_ROW(LID, LT)                                                                                   \
    case LID:                                                                                             \
        switch (rt) {                                                                                     \
            case EC_U8: return launch_masked_pair<LT, uint8_t, OP>(l, lm, r, rm, n, out, om, s);          \
            case EC_U16: return launch_masked_pair<LT, uint16_t, OP>(l, lm, r, rm, n, out, om, s);        \
            case EC_U32: return launch_masked_pair<LT, uint32_t, OP>(l, lm, r, rm, n, out, om, s);        \
            case EC_U64: return launch_masked_pair<LT, uint64_t, OP>(l, lm, r, rm, n, out, om, s);        \
            case EC_I8: return launch_masked_pair<LT, int8_t, OP>(l, lm, r, rm, n, out, om, s);           \
            case EC_I16: return launch_masked_pair<LT, int16_t, OP>(l, lm, r, rm, n, out, om, s);         \
            case EC_I32: return launch_masked_pair<LT, int32_t, OP>(l, lm, r, rm, n, out, om, s);         \
            case EC_I64: return launch_masked_pair<LT, int64_t, OP>(l, lm, r, rm, n, out, om, s);         \
            case EC_F32: return launch_masked_pair<LT, float, OP>(l, lm, r, rm, n, out, om, s);           \
            case EC_F64: return launch_masked_pair<LT, double, OP>(l, lm, r, rm, n, out, om, s);          \
        }                                                                                                 \
        break;
    switch (lt) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "masked_binop: bad dtype");
}

template <int OP>
ec_status dispatch_binop_scalar(int lt, const void* l, double rhs, size_t n, double* out, hipStream_t s) {
#define EC_ROW(LID, LT) case LID: return launch_scalar<LT, OP>(l, rhs, n, out, s);
    switch (lt) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "binop_scalar: bad dtype");
}

}  // namespace ecd

#ifdef EC_TU_OP
namespace ecd {
template ec_status dispatch_binop<EC_TU_OP>(int, const void*, int, const void*, size_t, double*, hipStream_t);
template ec_status dispatch_masked_binop<EC_TU_OP>(int, const void*, const uint8_t*, int, const void*, const uint8_t*,
                                                   size_t, double*, uint8_t*, hipStream_t);
template ec_status dispatch_binop_scalar<EC_TU_OP>(int, const void*, double, size_t, double*, hipStream_t);
}  // namespace ecd
#endif

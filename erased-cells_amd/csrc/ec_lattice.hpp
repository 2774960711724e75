// ec_lattice.hpp — CellType lattice as constexpr host functions.
// Restates CellType::{is_integral,is_signed,size_of,union,can_fit_into}
// (src/ctype.rs:55-131 of the reference); evaluated once per operation on the
// host instead of ≥4 times per cell.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "erased_cells.h"

namespace ecl {

constexpr bool valid(int t) { return t >= 0 && t < EC_NTYPES; }
constexpr bool is_integral(int t) { return t != EC_F32 && t != EC_F64; }          // ctype.rs:55-68
constexpr bool is_signed(int t) { return t >= EC_I8; }                             // ctype.rs:71-84 (floats signed)
constexpr size_t size_of(int t) {                                                  // ctype.rs:87-96
    constexpr size_t s[EC_NTYPES] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8};
    return valid(t) ? s[t] : 0;
}
constexpr size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

constexpr int union_of(int a, int b) {                                             // ctype.rs:99-126
    const size_t sa = size_of(a), sb = size_of(b);
    const bool ia = is_integral(a), ib = is_integral(b), ga = is_signed(a), gb = is_signed(b);
    size_t min_bytes = 0;
    if (ia && !ib) min_bytes = zmax(sb, 2 * sa);
    else if (!ia && ib) min_bytes = zmax(sa, 2 * sb);
    else if (ga && !gb) min_bytes = zmax(sa, 2 * sb);
    else if (!ga && gb) min_bytes = zmax(sb, 2 * sa);
    else min_bytes = zmax(sa, sb);
    const bool sg = ga || gb, in = ia && ib;
    if (in && min_bytes == 1) return sg ? EC_I8 : EC_U8;
    if (in && min_bytes == 2) return sg ? EC_I16 : EC_U16;
    if (in && min_bytes == 4) return sg ? EC_I32 : EC_U32;
    if (!in && min_bytes == 4) return EC_F32;
    if (in && min_bytes == 8) return sg ? EC_I64 : EC_U64;
    return EC_F64;
}

constexpr bool can_fit_into(int src, int dst) { return union_of(src, dst) == dst; }  // ctype.rs:129-131

constexpr int neg_result(int t) {                                                  // value.rs:224-240
    return t == EC_U8 ? EC_I16 : t == EC_U16 ? EC_I32 : (t == EC_U32 || t == EC_U64) ? EC_F64 : t;
}

template <typename T> struct dtype_of;
#define EC_DT(ID, T) template <> struct dtype_of<T> { static constexpr int value = ID; };
EC_DT(EC_U8, uint8_t) EC_DT(EC_U16, uint16_t) EC_DT(EC_U32, uint32_t) EC_DT(EC_U64, uint64_t)
EC_DT(EC_I8, int8_t) EC_DT(EC_I16, int16_t) EC_DT(EC_I32, int32_t) EC_DT(EC_I64, int64_t)
EC_DT(EC_F32, float) EC_DT(EC_F64, double)
#undef EC_DT

}  // namespace ecl

// ec_expr_jit.hip — the run-time compiled form of an expression program (see ec_expr_jit.hpp): source generator, hiprtc
// (resolved with dlopen), the cache of compiled programs, the background compile thread, module load and launch.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <iterator>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "ec_expr_jit.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

namespace ecd {

// ------------------------------------------------------------------------------------------------ the generator
namespace {

// Everything the generated kernel needs, self-contained (hiprtc sees no project header).  The word types are declared
// with alignment 1 for the same reason as ec_device.hpp's under_aligned: any cell offset runs the vector path.  1-byte
// cells travel as 16-bit words and are taken apart with shifts (hipcc drops `nt` from <N x i8> loads).
const char* const kPrelude = R"SRC(
typedef double D2 __attribute__((ext_vector_type(2)));
typedef unsigned int U2 __attribute__((ext_vector_type(2)));
typedef unsigned int U4 __attribute__((ext_vector_type(4)));
typedef unsigned short W1 __attribute__((aligned(1)));
typedef unsigned int W2 __attribute__((aligned(1)));
typedef U2 W4 __attribute__((aligned(1)));
typedef U4 W8 __attribute__((aligned(1)));
typedef D2 DO __attribute__((aligned(1)));
typedef unsigned long long u64;
// the result's 16-byte store: write-through and non-temporal, as the library's own value stores (ec_device.hpp stream_store_asm;
// `s_nop 1`: a VALU write of the data registers needs two wait states behind a store of more than 64 bits)
static __device__ __forceinline__ void st16(D2 v, DO* p) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v)); }
#define FOR _Pragma("unroll") for (int i = 0; i < N; ++i)
// cv_bin_op! on NaN (ec_device.hpp cell_op<OP, true>): the first NaN operand, quieted; the x86 default NaN when neither is one
static __device__ __forceinline__ double fixnan(double t, double a, double b) {
    const u64 f = (a != a) ? (__builtin_bit_cast(u64, a) | 0x0008000000000000ull)
                : (b != b) ? (__builtin_bit_cast(u64, b) | 0x0008000000000000ull) : 0xFFF8000000000000ull;
    return (t != t) ? __builtin_bit_cast(double, f) : t;
}
// The quotient of two integers of magnitude ≤ 131070 held as f64 (≤ 16-bit cells, their sums and differences): v_rcp_f64, one
// Newton step, the quotient, its exact residual and one correction — ec_device.hpp div_small_int, proven against the IEEE
// expansion on that whole square (tools/div_small_check.hip); b == 0 gives what the f64 divide gives: ±inf, the x86 default
// NaN for 0/0 (an integer-derived zero is +0).
static __device__ __forceinline__ double divs(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    const double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    const double z = a == 0.0 ? __builtin_bit_cast(double, 0xFFF8000000000000ull) : (a > 0.0 ? __builtin_inf() : -__builtin_inf());
    return b == 0.0 ? z : q;
}
// tested once per pair of cells (v_cmp_u_f64 r0, r1 is true when either is NaN), handled out of line, wave-uniform branch
#define NANFIX(A, B)                                                                                  \
    {                                                                                                 \
        bool nan = false;                                                                             \
        _Pragma("unroll") for (int i = 0; i + 1 < N; i += 2) nan = nan || __builtin_isunordered(t[i], t[i + 1]); \
        if (N & 1) nan = nan || (t[N - 1] != t[N - 1]);                                               \
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(nan) != 0, 0)) { FOR t[i] = fixnan(t[i], A, B); } \
    }
)SRC";

const char* const kCellType[10] = {"unsigned char", "unsigned short", "unsigned int", "unsigned long long", "signed char",
                                   "short",         "int",            "long long",    "float",              "double"};

void emit_loader(std::string& o, int k, int dt, bool nt) {
    const std::string K = std::to_string(k);
    auto LD = [&](const char* word) {
        return std::string(nt ? "__builtin_nontemporal_load(" : "*(") + "(const " + word + "*)b + pr)";
    };
    o += "static __device__ __forceinline__ void ld" + K + "(const char* b, unsigned long pr, double& c0, double& c1) {\n";
    switch (dt) {
        case EC_U8: o += "    const unsigned w = " + LD("W1") + ";\n    c0 = (double)(w & 0xffu); c1 = (double)(w >> 8);\n"; break;
        case EC_I8: o += "    const unsigned w = " + LD("W1") + ";\n    c0 = (double)((int)(w << 24) >> 24); c1 = (double)((int)(w << 16) >> 24);\n"; break;
        case EC_U16: o += "    const unsigned w = " + LD("W2") + ";\n    c0 = (double)(w & 0xffffu); c1 = (double)(w >> 16);\n"; break;
        case EC_I16: o += "    const unsigned w = " + LD("W2") + ";\n    c0 = (double)((int)(w << 16) >> 16); c1 = (double)((int)w >> 16);\n"; break;
        case EC_U32: o += "    const U2 v = " + LD("W4") + ";\n    const unsigned x = v.x, y = v.y;\n    c0 = (double)x; c1 = (double)y;\n"; break;
        case EC_I32: o += "    const U2 v = " + LD("W4") + ";\n    const unsigned x = v.x, y = v.y;\n    c0 = (double)(int)x; c1 = (double)(int)y;\n"; break;
        case EC_F32:
            o += "    const U2 v = " + LD("W4") + ";\n    const unsigned x = v.x, y = v.y;\n"
                 "    c0 = (double)__builtin_bit_cast(float, x); c1 = (double)__builtin_bit_cast(float, y);\n";
            break;
        default: {
            o += "    const U4 v = " + LD("W8") + ";\n    const unsigned x0 = v.x, x1 = v.y, y0 = v.z, y1 = v.w;\n"
                 "    const u64 x = ((u64)x1 << 32) | x0, y = ((u64)y1 << 32) | y0;\n";
            if (dt == EC_U64) o += "    c0 = (double)x; c1 = (double)y;\n";
            else if (dt == EC_I64) o += "    c0 = (double)(long long)x; c1 = (double)(long long)y;\n";
            else o += "    c0 = __builtin_bit_cast(double, x); c1 = __builtin_bit_cast(double, y);\n";
        }
    }
    o += "}\n";
    o += "static __device__ __forceinline__ double cell" + K + "(const char* p, unsigned long i) { return (double)__builtin_nontemporal_load((const " +
         kCellType[dt] + "*)p + i); }\n";
}

// pairs per lane per tile of the REDUCE variant (EC_EXPR_REDUCE_U for experiments).  Nothing is stored, so the loads in flight
// are all the memory-level parallelism there is.  Measured, NDVI / EVI statistics at 16384² with the full-tile path (every load
// of a tile issued before the first use): 2 pairs 0.413 / 0.527 ms, 4 pairs 0.378 / 0.497, 8 pairs 0.375 / 0.600
// (profiles/r03/expr_kernel.md).
int reduce_u() {
    static const int u = [] {
        const char* e = std::getenv("EC_EXPR_REDUCE_U");
        const int v = e ? std::atoi(e) : 4;
        return v == 1 || v == 2 || v == 4 || v == 8 ? v : 4;
    }();
    return u;
}

std::string operand(unsigned ref) {
    if (ref < unsigned(kRefReg0)) return "s" + std::to_string(ref) + "[i]";
    if (ref < unsigned(kRefScalar0)) return "r" + std::to_string(ref - kRefReg0) + "[i]";
    return "c" + std::to_string(ref - kRefScalar0);
}

}  // namespace

std::string expr_jit_source(const ExprArgs& ea, bool reduce) {
    const int ns = ea.nstreams;
    std::string o = "// generated by liberased_cells_hip (ec_expr_jit.hip): one expression program as straight-line code\n";
    o += kPrelude;
    for (int k = 0; k < ns; ++k) emit_loader(o, k, ea.dt[k], !((ea.cacheable >> k) & 1u));
    const char* scal = "double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7";
    o += "template <int N>\nstatic __device__ __forceinline__ void run(const double (&s0)[N], const double (&s1)[N], const double (&s2)[N], "
         "const double (&s3)[N], ";
    o += scal;
    o += ", double (&out)[N]) {\n    double r0[N], r1[N], r2[N], r3[N], t[N];\n";
    static const char* const kOp[4] = {"+", "-", "*", "/"};
    unsigned last = 0;
    // What the program PROVES about its values, used for one thing: a divide whose operands are both integers of magnitude
    // ≤ 131070 — cells of ≤ 16-bit integer streams (class 1) or a sum / difference of two such cells (class 2) — is the short
    // exact divide instead of the IEEE expansion (NDVI's divide; the interpreter cannot know, the generator reads the program).
    auto small_stream = [&](unsigned ref) { return ref < unsigned(ea.nstreams) && (ea.dt[ref] == EC_U8 || ea.dt[ref] == EC_I8 || ea.dt[ref] == EC_U16 || ea.dt[ref] == EC_I16); };
    int reg_class[kExprRegs] = {0, 0, 0, 0};
    auto small_operand = [&](unsigned ref) {
        if (ref < unsigned(kRefReg0)) return small_stream(ref);
        if (ref < unsigned(kRefScalar0)) return reg_class[ref - kRefReg0] != 0;
        return false;
    };
    for (int k = 0; k < ea.nsteps; ++k) {
        const unsigned step = static_cast<unsigned>(ea.prog[k >> 2] >> (16 * (k & 3))) & 0xffffu;
        const unsigned op = step & 3u, dst = (step >> 2) & 3u, a = (step >> 4) & 15u, b = (step >> 8) & 15u;
        const std::string A = operand(a), B = operand(b);
        if (op == unsigned(EC_DIV) && small_operand(a) && small_operand(b))
            o += "    FOR t[i] = divs(" + A + ", " + B + ");\n    FOR r" + std::to_string(dst) + "[i] = t[i];\n";
        else if (op != unsigned(EC_DIV) && small_operand(a) && small_operand(b))  // finite integers in, a finite number out: no NaN to fix
            o += "    FOR r" + std::to_string(dst) + "[i] = " + A + " " + kOp[op] + " " + B + ";\n";
        else
            o += "    FOR t[i] = " + A + " " + kOp[op] + " " + B + ";\n    NANFIX(" + A + ", " + B + ")\n    FOR r" + std::to_string(dst) + "[i] = t[i];\n";
        // class of the register written: a sum or difference of two ≤ 16-bit integer CELLS stays within ±131070
        reg_class[dst] = ((op == unsigned(EC_ADD) || op == unsigned(EC_SUB)) && a < unsigned(kRefReg0) && b < unsigned(kRefReg0) && small_stream(a) && small_stream(b)) ? 2 : 0;
        last = dst;
    }
    o += "    FOR out[i] = r" + std::to_string(last) + "[i];\n}\n";
    if (reduce) {
        // min / max of the valid cells' values, nothing stored: a grid-stride loop over the tiles (few workgroups, so that the
        // two atomics per workgroup at the end do not queue up), lane-local fold of the order keys, LDS tree, atomic max
        o += "static __device__ __forceinline__ long long okey(double d) {\n"
             "    const long long b = __builtin_bit_cast(long long, d);\n"
             "    return b ^ (long long)((unsigned long long)(b >> 63) >> 1);\n}\n";
        o += "extern \"C\" __global__ __launch_bounds__(256) void ec_expr_jit(const char* p0, const char* p1, const char* p2, const char* p3, ";
        o += scal;
        o += ", long long* __restrict__ keys2, unsigned long n, unsigned head, const char* m0, const char* m1, const char* m2, const char* m3, "
             "long long a0k, long long b0k) {\n"
             "    constexpr int U = " + std::to_string(reduce_u()) + ", NC = 2 * U;\n"
             "    const unsigned long npairs = (n - head) >> 1, TILE = 256ul * U, ntiles = (npairs + TILE - 1) / TILE;\n"
             "    long long a = a0k, b = b0k;  // {~key(min), key(max)} so far: the identities (f64::MAX, f64::MIN) of the reference's fold\n"
             "    double fmin = 1.7976931348623157e308, fmax = -1.7976931348623157e308;  // the same identities, by value\n";
        for (int k2 = 0; k2 < ns; ++k2)
            o += "    const char* b" + std::to_string(k2) + " = p" + std::to_string(k2) + " + (unsigned long)head * " + std::to_string(ecl::size_of(ea.dt[k2])) + ";\n";
        o += "    for (unsigned long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {\n"
             "        const unsigned long base = tile * TILE + threadIdx.x;\n"
             "        double s0[NC] = {}, s1[NC] = {}, s2[NC] = {}, s3[NC] = {}, o[NC];\n"
             "        bool valid[NC] = {};\n"
             "        const bool full = tile * TILE + TILE <= npairs;  // a full tile loads without guards: every load is issued before the first use\n";
        auto emit_tile_loads = [&](const char* ind) {
            for (int k2 = 0; k2 < ns; ++k2) {
                const std::string K = std::to_string(k2);
                o += std::string(ind) + "ld" + K + "(b" + K + ", pr, s" + K + "[2 * j], s" + K + "[2 * j + 1]);\n";
            }
            if (ea.nmask > 0) {
                o += std::string(ind) + "unsigned mk = 0xffffu;\n";
                for (int k2 = 0; k2 < ea.nmask; ++k2) {
                    const bool nt = !((ea.cacheable >> (4 + k2)) & 1u);
                    o += std::string(ind) + "mk &= " + (nt ? "__builtin_nontemporal_load(" : "*(") + "(const W1*)(m" + std::to_string(k2) + " + head) + pr);\n";
                }
                o += std::string(ind) + "valid[2 * j] = (mk & 0xffu) != 0; valid[2 * j + 1] = (mk >> 8) != 0;\n";
            } else {
                o += std::string(ind) + "valid[2 * j] = valid[2 * j + 1] = true;\n";
            }
        };
        o += "        if (full) {\n            _Pragma(\"unroll\") for (int j = 0; j < U; ++j) {\n                const unsigned long pr = base + j * 256ul;\n";
        emit_tile_loads("                ");
        o += "            }\n        } else {\n            _Pragma(\"unroll\") for (int j = 0; j < U; ++j) {\n                const unsigned long pr = base + j * 256ul;\n"
             "                if (pr < npairs) {\n";
        emit_tile_loads("                    ");
        o += "                }\n            }\n        }\n";
        o += "        run<NC>(s0, s1, s2, s3, c0, c1, c2, c3, c4, c5, c6, c7, o);\n"
             // a tile without a NaN or a zero among its valid values (the common case) is folded by VALUE — v_min_f64 / v_max_f64,
             // 3 instructions per cell instead of the 13 of the key fold: by value and by total_cmp agree when no two bit patterns
             // compare equal and nothing is unordered
             "        bool special = false;\n"
             "        _Pragma(\"unroll\") for (int i = 0; i < NC; ++i) special = special || (valid[i] && __builtin_amdgcn_class(o[i], 0x63));\n"
             "        if (__builtin_amdgcn_ballot_w64(special) == 0) {\n"
             "            _Pragma(\"unroll\") for (int i = 0; i < NC; ++i) {\n"
             "                fmin = __builtin_fmin(fmin, valid[i] ? o[i] : fmin);\n                fmax = __builtin_fmax(fmax, valid[i] ? o[i] : fmax);\n            }\n"
             "        } else {\n"
             "            _Pragma(\"unroll\") for (int i = 0; i < NC; ++i)\n                if (valid[i]) {\n"
             "                    const long long k = okey(o[i]);\n                    a = ~k > a ? ~k : a;\n                    b = k > b ? k : b;\n                }\n"
             "        }\n"
             "    }\n"
             "    {\n        const long long k0 = okey(fmin), k1 = okey(fmax);\n        a = ~k0 > a ? ~k0 : a;\n        b = k1 > b ? k1 : b;\n    }\n"
             "    if (blockIdx.x == 0 && threadIdx.x < 2) {  // the peeled head cell (lane 0) and the odd tail cell (lane 1)\n"
             "        const bool do_it = threadIdx.x == 0 ? head != 0 : ((n - head) & 1) != 0;\n"
             "        const unsigned long i = threadIdx.x == 0 ? 0 : n - 1;\n"
             "        if (do_it) {\n";
        for (int k2 = 0; k2 < 4; ++k2) {
            const std::string K = std::to_string(k2);
            o += "            const double a" + K + "[1] = {" + (k2 < ns ? "cell" + K + "(p" + K + ", i)" : std::string("0.0")) + "};\n";
        }
        o += "            double q[1];\n            run<1>(a0, a1, a2, a3, c0, c1, c2, c3, c4, c5, c6, c7, q);\n            bool ok = true;\n";
        for (int k2 = 0; k2 < ea.nmask; ++k2)
            o += "            ok = ok && __builtin_nontemporal_load((const unsigned char*)m" + std::to_string(k2) + " + i) != 0;\n";
        o += "            if (ok) {\n                const long long k = okey(q[0]);\n                a = ~k > a ? ~k : a;\n                b = k > b ? k : b;\n            }\n"
             "        }\n    }\n"
             "    __shared__ long long sa[256], sb[256];\n"
             "    sa[threadIdx.x] = a;\n    sb[threadIdx.x] = b;\n    __syncthreads();\n"
             "    for (int w = 128; w > 0; w >>= 1) {\n"
             "        if ((int)threadIdx.x < w) {\n"
             "            sa[threadIdx.x] = sa[threadIdx.x + w] > sa[threadIdx.x] ? sa[threadIdx.x + w] : sa[threadIdx.x];\n"
             "            sb[threadIdx.x] = sb[threadIdx.x + w] > sb[threadIdx.x] ? sb[threadIdx.x + w] : sb[threadIdx.x];\n"
             "        }\n        __syncthreads();\n    }\n"
             "    if (threadIdx.x == 0) {\n"
             "        __hip_atomic_fetch_max(keys2, sa[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
             "        __hip_atomic_fetch_max(keys2 + 1, sb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
             "    }\n}\n";
        return o;
    }
    o += "extern \"C\" __global__ __launch_bounds__(256) void ec_expr_jit(const char* p0, const char* p1, const char* p2, const char* p3, ";
    o += scal;
    o += ", double* __restrict__ out, unsigned long n, unsigned head, const char* m0, const char* m1, const char* m2, const char* m3, "
         "unsigned char* __restrict__ out_mask) {\n"
         "    constexpr int U = 2, NC = 2 * U;\n"
         "    const unsigned long npairs = (n - head) >> 1, TILE = 256ul * U;\n"
         "    const unsigned long blk = blockIdx.x, tile = (blk & 1) ? (unsigned long)gridDim.x - 1 - (blk >> 1) : (blk >> 1);  // two fronts\n"
         "    const unsigned long base = tile * TILE + threadIdx.x;\n"
         "    DO* op = (DO*)(out + head);\n"
         "    double s0[NC] = {}, s1[NC] = {}, s2[NC] = {}, s3[NC] = {}, o[NC];\n";
    for (int k = 0; k < ns; ++k)
        o += "    const char* b" + std::to_string(k) + " = p" + std::to_string(k) + " + (unsigned long)head * " + std::to_string(ecl::size_of(ea.dt[k])) + ";\n";
    // a full tile loads without guards (every load of the tile is issued before the first use); the last tile guards each pair
    o += "    const bool full = tile * TILE + TILE <= npairs;\n"
         "    if (full) {\n        _Pragma(\"unroll\") for (int j = 0; j < U; ++j) {\n            const unsigned long pr = base + j * 256ul;\n";
    for (int k = 0; k < ns; ++k) {
        const std::string K = std::to_string(k);
        o += "            ld" + K + "(b" + K + ", pr, s" + K + "[2 * j], s" + K + "[2 * j + 1]);\n";
    }
    o += "        }\n    } else {\n        _Pragma(\"unroll\") for (int j = 0; j < U; ++j) {\n            const unsigned long pr = base + j * 256ul;\n            if (pr < npairs) {\n";
    for (int k = 0; k < ns; ++k) {
        const std::string K = std::to_string(k);
        o += "                ld" + K + "(b" + K + ", pr, s" + K + "[2 * j], s" + K + "[2 * j + 1]);\n";
    }
    o += "            }\n        }\n    }\n"
         "    run<NC>(s0, s1, s2, s3, c0, c1, c2, c3, c4, c5, c6, c7, o);\n"
         "    _Pragma(\"unroll\") for (int j = 0; j < U; ++j) {\n        const unsigned long pr = base + j * 256ul;\n"
         "        if (full || pr < npairs) st16(D2{o[2 * j], o[2 * j + 1]}, op + pr);\n    }\n"
         "    if (blockIdx.x == 0 && threadIdx.x < 2) {  // the peeled head cell (lane 0) and the odd tail cell (lane 1)\n"
         "        const bool do_it = threadIdx.x == 0 ? head != 0 : ((n - head) & 1) != 0;\n"
         "        const unsigned long i = threadIdx.x == 0 ? 0 : n - 1;\n"
         "        if (do_it) {\n";
    for (int k = 0; k < 4; ++k) {
        const std::string K = std::to_string(k);
        o += "            const double a" + K + "[1] = {" + (k < ns ? "cell" + K + "(p" + K + ", i)" : std::string("0.0")) + "};\n";
    }
    o += "            double q[1];\n            run<1>(a0, a1, a2, a3, c0, c1, c2, c3, c4, c5, c6, c7, q);\n"
         "            __builtin_nontemporal_store(q[0], out + i);\n        }\n    }\n";
    if (ea.nmask > 0) {  // the AND of the distinct masks (ec_expr.hpp expr_mask_phase), 16 mask bytes per lane, its own lane map
        o += "    {\n        const unsigned long ngroups = n / 16, stride = (unsigned long)gridDim.x * 256ul;\n"
             "        for (unsigned long g = (unsigned long)blockIdx.x * 256ul + threadIdx.x; g < ngroups; g += stride) {\n"
             "            U4 acc = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};\n";
        for (int k = 0; k < ea.nmask; ++k) {
            const bool nt = !((ea.cacheable >> (4 + k)) & 1u);
            o += std::string("            acc &= ") + (nt ? "__builtin_nontemporal_load(" : "*(") + "(const W8*)m" + std::to_string(k) + " + g);\n";
        }
        o += "            __builtin_nontemporal_store(acc, (W8*)out_mask + g);\n        }\n"
             "        if (blockIdx.x == 0)\n            for (unsigned long i = ngroups * 16 + threadIdx.x; i < n; i += 256ul) {\n"
             "                unsigned char acc = __builtin_nontemporal_load((const unsigned char*)m0 + i);\n";
        for (int k = 1; k < ea.nmask; ++k)
            o += "                acc &= __builtin_nontemporal_load((const unsigned char*)m" + std::to_string(k) + " + i);\n";
        o += "                __builtin_nontemporal_store(acc, out_mask + i);\n            }\n    }\n";
    }
    o += "}\n";
    return o;
}

// ------------------------------------------------------------------------------------------------ hiprtc, lazily
namespace {

struct Hiprtc {
    int (*create)(void** prog, const char* src, const char* name, int nh, const char** headers, const char** names) = nullptr;
    int (*compile)(void* prog, int nopt, const char** opts) = nullptr;
    int (*code_size)(void* prog, size_t* n) = nullptr;
    int (*code)(void* prog, char* out) = nullptr;
    int (*log_size)(void* prog, size_t* n) = nullptr;
    int (*log)(void* prog, char* out) = nullptr;
    int (*destroy)(void** prog) = nullptr;
    const char* (*error_string)(int) = nullptr;
    std::string load_error;
};

Hiprtc g_rtc;
std::once_flag g_rtc_once;

void load_hiprtc() {
    void* h = nullptr;
    if (const char* named = std::getenv("EC_HIPRTC_LIB")) {  // another ROCm installation's hiprtc, or none at all (tests)
        h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    }
    if (!h) {
        const char* e = dlerror();
        g_rtc.load_error = std::string("libhiprtc not found: ") + (e ? e : "dlopen failed");
        return;
    }
#define EC_SYM(field, name) g_rtc.field = reinterpret_cast<decltype(g_rtc.field)>(dlsym(h, name))
    EC_SYM(create, "hiprtcCreateProgram");
    EC_SYM(compile, "hiprtcCompileProgram");
    EC_SYM(code_size, "hiprtcGetCodeSize");
    EC_SYM(code, "hiprtcGetCode");
    EC_SYM(log_size, "hiprtcGetProgramLogSize");
    EC_SYM(log, "hiprtcGetProgramLog");
    EC_SYM(destroy, "hiprtcDestroyProgram");
    EC_SYM(error_string, "hiprtcGetErrorString");
#undef EC_SYM
    if (!g_rtc.create || !g_rtc.compile || !g_rtc.code_size || !g_rtc.code || !g_rtc.destroy) g_rtc.load_error = "libhiprtc lacks an expected symbol";
}

const Hiprtc* hiprtc() {
    std::call_once(g_rtc_once, load_hiprtc);
    return g_rtc.load_error.empty() ? &g_rtc : nullptr;
}

}  // namespace

ec_status expr_jit_compile(const std::string& source, const std::string& arch, std::string* code, std::string* log) {
    // the one processor this library is built for (hiprtc crashes on a name it does not know instead of refusing it)
    if (arch != "gfx950") return set_error(EC_ERR_ARG, "expression compile: processor '%s' (this library targets gfx950)", arch.c_str());
    const Hiprtc* R = hiprtc();
    if (!R) return set_error_text(EC_ERR_HIP, "expression compile: " + g_rtc.load_error);
    // one compile at a time, process-wide: the background thread, callers in expr_jit = 2 and ec_expr_source may all arrive
    // here at once, and two concurrent hiprtc compiles of different programs took the process down ("pure virtual method
    // called") in a soak of the round's entry points from five host threads
    static std::mutex one_at_a_time;
    std::lock_guard<std::mutex> lk(one_at_a_time);
    void* prog = nullptr;
    int rc = R->create(&prog, source.c_str(), "ec_expr_jit.hip", 0, nullptr, nullptr);
    if (rc != 0) return set_error(EC_ERR_HIP, "hiprtcCreateProgram: %s", R->error_string ? R->error_string(rc) : "failed");
    const std::string arch_opt = "--offload-arch=" + arch;
    // the arithmetic must be the library's: no contraction into FMAs, IEEE divide and no fast-math (the csrc Makefile's flags)
    const char* opts[] = {arch_opt.c_str(), "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17"};
    rc = R->compile(prog, 5, opts);
    size_t ls = 0;
    if (log && R->log_size && R->log && R->log_size(prog, &ls) == 0 && ls > 1) {
        log->assign(ls, '\0');
        (void)R->log(prog, &(*log)[0]);
    }
    if (rc != 0) {
        const std::string why = R->error_string ? R->error_string(rc) : "failed";
        (void)R->destroy(&prog);
        return set_error_text(EC_ERR_HIP, "hiprtcCompileProgram: " + why + (log && !log->empty() ? "\n" + *log : std::string()));
    }
    size_t cs = 0;
    rc = R->code_size(prog, &cs);
    if (rc == 0 && cs > 0) {
        code->assign(cs, '\0');
        rc = R->code(prog, &(*code)[0]);
    }
    (void)R->destroy(&prog);
    if (rc != 0 || cs == 0) return set_error(EC_ERR_HIP, "hiprtcGetCode failed");
    return EC_OK;
}

// ------------------------------------------------------------------------------------------------ the cache
namespace {

enum : int { kNew = 0, kCompiling = 1, kReady = 2, kFailed = 3 };
constexpr size_t kMaxPrograms = 4096;  // cache entries (a code object is ≈ 10 KB)

struct Entry {
    std::atomic<int> state{kNew};
    std::atomic<int64_t> work{0};  // cell-steps interpreted so far (the trigger of the background compile)
    std::string source, arch, code, error;
    std::mutex mu;                              // guards fn / modules
    std::map<int, hipFunction_t> fn;            // per device
    std::vector<std::pair<int, hipModule_t>> modules;
};

std::mutex g_mu;  // guards g_cache and the queue
std::unordered_map<std::string, std::shared_ptr<Entry>> g_cache;
std::deque<std::shared_ptr<Entry>> g_queue;
std::condition_variable g_cv;
bool g_stop = false;
std::atomic<int64_t> g_compiles{0}, g_failures{0}, g_jit_launches{0};

void compile_entry(Entry& e) {
    std::string log;
    const ec_status st = expr_jit_compile(e.source, e.arch, &e.code, &log);
    if (st == EC_OK) {
        g_compiles.fetch_add(1, std::memory_order_relaxed);
        e.state.store(kReady, std::memory_order_release);
    } else {
        e.error = last_error_text();
        g_failures.fetch_add(1, std::memory_order_relaxed);
        e.state.store(kFailed, std::memory_order_release);
    }
}

// The background compiler: one thread, started with the first request, joined when the library is unloaded.
struct Compiler {
    std::thread th;
    void ensure() {
        if (!th.joinable()) th = std::thread([] {
            for (;;) {
                std::shared_ptr<Entry> e;
                {
                    std::unique_lock<std::mutex> lk(g_mu);
                    g_cv.wait(lk, [] { return g_stop || !g_queue.empty(); });
                    if (g_stop) return;
                    e = std::move(g_queue.front());
                    g_queue.pop_front();
                }
                compile_entry(*e);
            }
        });
    }
    // Stop and drain: the thread finishes the compile it is in (hiprtc offers no way to abandon one), programs still queued
    // go back to "never compiled" and are queued again when they have run long enough.  Called by ec_shutdown (expr_jit_release),
    // so that by the time the process runs static destructors — when the dlopen'ed libhiprtc may already be finalised — there
    // is no compile in flight and nothing left to join; a later ec_init starts a fresh thread with the next request.
    void stop() {
        {
            std::lock_guard<std::mutex> lk(g_mu);
            g_stop = true;
        }
        g_cv.notify_all();
        if (th.joinable()) th.join();
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& e : g_queue) {
            e->work.store(0, std::memory_order_relaxed);
            e->state.store(kNew, std::memory_order_release);
        }
        g_queue.clear();
        g_stop = false;
    }
    ~Compiler() { stop(); }
};
Compiler g_compiler;

std::string key_of(const ExprArgs& ea, const std::string& arch, bool reduce) {
    std::string k(reinterpret_cast<const char*>(ea.prog), sizeof ea.prog);
    k.append(reinterpret_cast<const char*>(ea.dt), sizeof ea.dt);
    k.push_back(static_cast<char>(ea.nstreams));
    k.push_back(static_cast<char>(ea.nsteps));
    k.push_back(static_cast<char>(ea.cacheable & ((1u << ea.nstreams) - 1u)));
    k.push_back(static_cast<char>(ea.nmask));
    k.push_back(static_cast<char>((ea.cacheable >> 4) & ((1u << ea.nmask) - 1u)));
    k.push_back(reduce ? 'r' : 's');
    return k + arch;
}

// gcnArchName is "gfx950:sramecc+:xnack-": the processor alone makes a code object any feature setting loads
ec_status device_arch(int dev, std::string* arch) {
    static std::mutex mu;
    static std::map<int, std::string> known;
    std::lock_guard<std::mutex> lk(mu);
    auto it = known.find(dev);
    if (it == known.end()) {
        hipDeviceProp_t prop;
        ec_status st = check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
        if (st != EC_OK) return st;
        std::string a = prop.gcnArchName;
        it = known.emplace(dev, a.substr(0, a.find(':'))).first;
    }
    *arch = it->second;
    return EC_OK;
}

}  // namespace

ec_status expr_jit_launch(const ExprArgs& ea, size_t n, double* out, uint8_t* out_mask, int64_t* keys2, hipStream_t s, bool* launched) {
    *launched = false;
    const bool reduce = keys2 != nullptr;
    const int mode = tuning().expr_jit.load();
    if (mode == 0) return EC_OK;
    const int dev = current_device();
    std::string arch;
    ec_status st = device_arch(dev, &arch);
    if (st != EC_OK) return st;
    if (arch != "gfx950") return EC_OK;  // the compiled form is written for gfx950 like the rest of the library
    std::shared_ptr<Entry> e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const std::string key = key_of(ea, arch, reduce);
        auto it = g_cache.find(key);
        if (it == g_cache.end()) {
            if (g_cache.size() >= kMaxPrograms) {
                // a process that keeps inventing programs: forget those that never ran long enough to be compiled (they hold a
                // work counter only); if every entry is a compiled program, the new one is interpreted
                // — and that no caller holds at this moment (use_count 1: the cache's own reference; a caller takes its reference
                // under g_mu): an entry erased under a caller that goes on to compile it would load a module nobody unloads
                for (auto c = g_cache.begin(); c != g_cache.end();)
                    c = (c->second.use_count() == 1 && c->second->state.load(std::memory_order_acquire) == kNew) ? g_cache.erase(c) : std::next(c);
                if (g_cache.size() >= kMaxPrograms) return EC_OK;
            }
            it = g_cache.emplace(key, std::make_shared<Entry>()).first;
            it->second->arch = arch;
        }
        e = it->second;
    }
    int state = e->state.load(std::memory_order_acquire);
    if (state == kNew) {
        const int64_t work = e->work.fetch_add(static_cast<int64_t>(n) * ea.nsteps, std::memory_order_relaxed) + static_cast<int64_t>(n) * ea.nsteps;
        int expected = kNew;
        if (mode >= 2) {  // on the calling thread, now
            if (e->state.compare_exchange_strong(expected, kCompiling)) {
                e->source = expr_jit_source(ea, reduce);
                compile_entry(*e);
            } else {
                while (e->state.load(std::memory_order_acquire) == kCompiling) std::this_thread::yield();  // another caller compiles it
            }
            state = e->state.load(std::memory_order_acquire);
            if (state == kFailed) return set_error_text(EC_ERR_HIP, "ec_expr (expr_jit = 2): " + e->error);
        } else if (work >= (int64_t(1) << 31)) {
            if (e->state.compare_exchange_strong(expected, kCompiling)) {
                e->source = expr_jit_source(ea, reduce);
                std::lock_guard<std::mutex> lk(g_mu);
                g_compiler.ensure();
                g_queue.push_back(e);
                g_cv.notify_one();
            }
            return EC_OK;
        } else {
            return EC_OK;
        }
    }
    if (state != kReady) return EC_OK;  // compiling or failed: interpret
    hipFunction_t fn = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        auto it = e->fn.find(dev);
        if (it == e->fn.end()) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return EC_OK;  // no module load inside a capture
            hipModule_t mod = nullptr;
            st = check_hip(hipModuleLoadData(&mod, e->code.data()), "hipModuleLoadData(expression kernel)");
            if (st != EC_OK) return st;
            st = check_hip(hipModuleGetFunction(&fn, mod, "ec_expr_jit"), "hipModuleGetFunction(ec_expr_jit)");
            if (st != EC_OK) {
                (void)hipModuleUnload(mod);
                return st;
            }
            e->modules.emplace_back(dev, mod);
            e->fn.emplace(dev, fn);
        } else {
            fn = it->second;
        }
    }
    const char* p[4] = {static_cast<const char*>(ea.p[0]), static_cast<const char*>(ea.p[1]), static_cast<const char*>(ea.p[2]),
                        static_cast<const char*>(ea.p[3])};
    double c[8];
    for (int k = 0; k < 8; ++k) c[k] = ea.sc[k];
    unsigned long nn = n;
    unsigned head = ea.head;
    const char* m[4] = {reinterpret_cast<const char*>(ea.m[0]), reinterpret_cast<const char*>(ea.m[1]), reinterpret_cast<const char*>(ea.m[2]),
                        reinterpret_cast<const char*>(ea.m[3])};
    const size_t npairs = (n - head) >> 1;
    if (reduce) {
        long long a0k = ~order_key<double>(std::numeric_limits<double>::max()), b0k = order_key<double>(std::numeric_limits<double>::lowest());
        long long* k2 = reinterpret_cast<long long*>(keys2);
        void* params[] = {&p[0], &p[1], &p[2], &p[3], &c[0], &c[1], &c[2], &c[3], &c[4], &c[5], &c[6], &c[7], &k2, &nn, &head,
                          &m[0], &m[1], &m[2], &m[3], &a0k, &b0k};
        const size_t per_tile = size_t(256) * reduce_u();
        const size_t tiles = (npairs + per_tile - 1) / per_tile;
        const unsigned grid = static_cast<unsigned>(std::min<size_t>(std::max<size_t>(tiles, 1), size_t(device_cus()) * 8));
        st = check_hip(hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, s, params, nullptr), "hipModuleLaunchKernel(ec_expr_jit, reduce)");
    } else {
        void* params[] = {&p[0], &p[1], &p[2], &p[3], &c[0], &c[1], &c[2], &c[3], &c[4], &c[5], &c[6], &c[7], &out, &nn, &head,
                          &m[0], &m[1], &m[2], &m[3], &out_mask};
        const unsigned grid = grid_for((npairs + 511) / 512);
        st = check_hip(hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, s, params, nullptr), "hipModuleLaunchKernel(ec_expr_jit)");
    }
    if (st != EC_OK) return st;
    g_jit_launches.fetch_add(1, std::memory_order_relaxed);
    *launched = true;
    return EC_OK;
}

// ec_shutdown: unload the modules (the compiled code objects stay cached and are loaded again on demand)
void expr_jit_release() {
    g_compiler.stop();  // no compile in flight past this point
    std::vector<std::shared_ptr<Entry>> all;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_cache) all.push_back(kv.second);
    }
    for (auto& e : all) {
        std::lock_guard<std::mutex> lk(e->mu);
        for (auto& dm : e->modules)
            if (hipSetDevice(dm.first) == hipSuccess) (void)hipModuleUnload(dm.second);
        e->modules.clear();
        e->fn.clear();
    }
}

int64_t expr_jit_stat(const char* key, bool* known) {
    *known = true;
    if (!std::strcmp(key, "expr_jit_compiles")) return g_compiles.load(std::memory_order_relaxed);
    if (!std::strcmp(key, "expr_jit_failures")) return g_failures.load(std::memory_order_relaxed);
    if (!std::strcmp(key, "expr_jit_launches")) return g_jit_launches.load(std::memory_order_relaxed);
    if (!std::strcmp(key, "expr_jit_programs")) {
        std::lock_guard<std::mutex> lk(g_mu);
        return static_cast<int64_t>(g_cache.size());
    }
    *known = false;
    return 0;
}

}  // namespace ecd

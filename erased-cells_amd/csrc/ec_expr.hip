// ec_expr.hip — ABI entry points of the expression-program kernel (ec_expr.hpp): validation of a program, register
// liveness, operand set-up and dispatch by the streams' byte widths.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include <limits>
#include <mutex>
#include <string>

#include "ec_expr.hpp"
#include "ec_expr_fixed.hpp"
#include "ec_expr_jit.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

using namespace ecd;

static std::atomic<int64_t> g_interpreted{0};  // launches of the interpreter's vector kernel (ec_stat_get("expr_interp_launches"))

namespace ecd {
int64_t expr_stat(const char* key, bool* known) {
    if (!std::strcmp(key, "expr_interp_launches")) {
        *known = true;
        return g_interpreted.load(std::memory_order_relaxed);
    }
    const int64_t v = expr_fixed_stat(key, known);  // expr_fixed_launches
    if (*known) return v;
    return expr_jit_stat(key, known);
}
}  // namespace ecd

// The program part of a call, checked and packed: counts, cell types, steps (with the host's marks), no pointers.
static ec_status program_of(ExprArgs& ea, const ec_dtype* dt, int32_t n_streams, int32_t n_scalars, const ec_expr_step* steps,
                            int32_t n_steps, const char* what) {
    if (n_streams < 1 || n_streams > kExprMaxStreams) return set_error(EC_ERR_ARG, "%s: %d operand streams (1..%d)", what, int(n_streams), kExprMaxStreams);
    if (n_scalars < 0 || n_scalars > kExprMaxScalars) return set_error(EC_ERR_ARG, "%s: %d scalars (0..%d)", what, int(n_scalars), kExprMaxScalars);
    if (n_steps < 1 || n_steps > kExprMaxSteps) return set_error(EC_ERR_ARG, "%s: %d steps (1..%d)", what, int(n_steps), kExprMaxSteps);
    if (!dt || !steps) return set_error(EC_ERR_ARG, "%s: null pointer", what);
    for (int k = 0; k < n_streams; ++k) {
        if (!ecl::valid(dt[k])) return set_error(EC_ERR_UNSUPPORTED_TYPE, "%s: bad dtype of stream %d", what, k);
        ea.dt[k] = static_cast<int8_t>(dt[k]);
    }
    bool written[kExprRegs] = {false, false, false, false};
    for (int k = 0; k < n_steps; ++k) {
        const ec_expr_step& st = steps[k];
        if (st.op < EC_ADD || st.op > EC_DIV) return set_error(EC_ERR_ARG, "%s: step %d: bad op", what, k);
        if (st.dst < 0 || st.dst >= kExprRegs) return set_error(EC_ERR_ARG, "%s: step %d: destination register %d (0..%d)", what, k, int(st.dst), kExprRegs - 1);
        for (int side = 0; side < 2; ++side) {
            const int ref = side ? st.b : st.a;
            bool ok = false;
            if (ref >= kRefStream0 && ref < kRefReg0) ok = ref - kRefStream0 < n_streams;
            else if (ref >= kRefReg0 && ref < kRefScalar0) ok = written[ref - kRefReg0];  // a register is read only after a step wrote it
            else if (ref >= kRefScalar0 && ref < kRefEnd) ok = ref - kRefScalar0 < n_scalars;
            if (!ok) return set_error(EC_ERR_ARG, "%s: step %d: operand %c refers to %d — no such stream / scalar, or a register no earlier step wrote",
                                      what, k, side ? 'b' : 'a', ref);
        }
        written[st.dst] = true;
    }
    // Marks for the kernel (ec_expr.hpp expr_run): an operand that names the register the previous step wrote is read from
    // the accumulator; a result that only such an operand (or nothing) reads before the register is overwritten is not filed.
    for (int k = 0; k < n_steps; ++k) {
        const ec_expr_step& st = steps[k];
        uint64_t w = uint64_t(st.op) | uint64_t(st.dst) << 2 | uint64_t(st.a) << 4 | uint64_t(st.b) << 8;
        if (k > 0) {
            const int prev = kRefReg0 + steps[k - 1].dst;
            if (st.a == prev) w |= kStepAAcc;
            if (st.b == prev) w |= kStepBAcc;
        }
        bool read_later = false;
        for (int j = k + 1; j < n_steps && !read_later; ++j) {
            const int me = kRefReg0 + st.dst;
            if (j > k + 1 && (steps[j].a == me || steps[j].b == me)) read_later = true;
            if (steps[j].dst == st.dst) break;  // overwritten (after step j's own reads)
        }
        if (!read_later) w |= kStepNoFile;
        ea.prog[k >> 2] |= w << (16 * (k & 3));
    }
    ea.nstreams = static_cast<int8_t>(n_streams);
    ea.nsteps = static_cast<int8_t>(n_steps);
    return EC_OK;
}

// Everything of a call but its outputs: the program, the streams, the scalars, the distinct masks, the peel and the load policy.
// `*aligned`: the vector kernels may run on these pointers (always, unless the unaligned_vector knob is off).
static ec_status prepare_expr(ExprArgs& ea, bool* aligned_out, int* cls, const ec_dtype* dt, const void* const* p, int32_t n_streams,
                              const uint8_t* const* masks, const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps,
                              int32_t n_steps, size_t n, const char* what) {
    ec_status pst = program_of(ea, dt, n_streams, n_scalars, steps, n_steps, what);
    if (pst != EC_OK) return pst;
    if (!p || (n_scalars > 0 && !scalars)) return set_error(EC_ERR_ARG, "%s: null pointer", what);
    bool aligned = true;
    for (int k = 0; k < n_streams; ++k) {
        if (!p[k]) return set_error(EC_ERR_ARG, "%s: stream %d is null", what, k);
        ea.p[k] = p[k];
        aligned = aligned && aligned_to(p[k], 16);
    }
    for (int k = n_streams; k < kExprMaxStreams; ++k) ea.p[k] = p[0];  // never read (class 0)
    for (int k = 0; k < n_scalars; ++k) {
        if (!ecl::valid(scalars[k].dtype)) return set_error(EC_ERR_ARG, "%s: bad dtype of scalar %d", what, k);
        ea.sc[k] = ec_value_to_f64(&scalars[k]);  // impl $trt<R: Into<CellValue>> (src/buffer.rs:346-352): widened once, here
    }
    if (masks) {
        for (int k = 0; k < n_streams; ++k) {
            if (!masks[k]) return set_error(EC_ERR_ARG, "%s: null mask %d", what, k);
            bool seen = false;
            for (int j = 0; j < ea.nmask; ++j) seen = seen || ea.m[j] == masks[k];
            if (!seen) {
                ea.m[ea.nmask++] = masks[k];
                aligned = aligned && aligned_to(masks[k], 16);
            }
        }
    }
    *aligned_out = aligned;
    // peel one leading cell when that puts more of the 1-byte streams on even addresses (peel_head's rule)
    unsigned c0 = 0, c1 = 0;
    size_t stream_bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < n_streams; ++k) {
        const size_t bytes = ecl::size_of(dt[k]);
        c0 += peel_cost(p[k], bytes, 0);
        c1 += peel_cost(p[k], bytes, 1);
        cls[k] = fused_class_index(bytes);
        stream_bytes[k] = n * bytes;
        for (int j = 0; j < k; ++j)
            if (p[j] == p[k]) stream_bytes[k] = 0;  // the same buffer again: its bytes count once (policy copied below)
    }
    for (int j = 0; j < ea.nmask; ++j) stream_bytes[4 + j] = n;
    ea.head = (n >= 2 && tuning().peel && c1 < c0) ? 1 : 0;
    unsigned policy = cache_plan(stream_bytes, 8, n * sizeof(double));
    for (int k = 1; k < n_streams; ++k)
        for (int j = 0; j < k; ++j)
            if (p[j] == p[k]) policy = (policy & ~(1u << k)) | (((policy >> j) & 1u) << k);  // one buffer, one policy
    ea.cacheable = static_cast<uint8_t>(policy);
    return EC_OK;
}

static ec_status launch_expr(const ec_dtype* dt, const void* const* p, int32_t n_streams, const uint8_t* const* masks,
                             const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, size_t n,
                             double* out, uint8_t* out_mask, hipStream_t s, const char* what) {
    ExprArgs ea{};
    bool aligned = true;
    int cls[kExprMaxStreams] = {0, 0, 0, 0};
    if (!out) return set_error(EC_ERR_ARG, "%s: null pointer", what);
    if (masks && !out_mask) return set_error(EC_ERR_ARG, "%s: null out_mask", what);
    ec_status pst = prepare_expr(ea, &aligned, cls, dt, p, n_streams, masks, scalars, n_scalars, steps, n_steps, n, what);
    if (pst != EC_OK) return pst;
    aligned = aligned && aligned_to(out, 16) && (!masks || aligned_to(out_mask, 16));
    if (!aligned) {
        ea.head = 0;
        k_expr_cellwise<0><<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(ea, out, out_mask, n);
        return check_launch("expr(cellwise)");
    }
    // A program of the ahead-of-time catalogue (ec_expr_fixed.hpp: NDVI, (a + b) * c, EVI, a * k0 + k1 over streams of one cell
    // width) runs as its built-in straight-line kernel: from its first launch, inside a stream capture, without libhiprtc.
    bool fixed = false;
    ec_status fst = expr_fixed_launch(ea, n, out, out_mask, s, &fixed);
    if (fst != EC_OK) return fst;
    if (fixed) return check_launch("expr(ahead-of-time)");
    // The program compiled for itself (ec_expr_jit.hpp), once it is ready: values and the masks' AND as straight-line code in
    // one launch.  Until then — and whenever expr_jit is 0 — the interpreter below.
    bool compiled = false;
    ec_status jst = expr_jit_launch(ea, n, out, out_mask, nullptr, s, &compiled);
    if (jst != EC_OK) return jst;
    if (compiled) return check_launch("expr(compiled)");
    g_interpreted.fetch_add(1, std::memory_order_relaxed);
    const size_t per_tile = size_t(kBlock) * kExprU;
    const unsigned grid = grid_for((((n - ea.head) >> 1) + per_tile - 1) / per_tile);
    ExprKernel kern = nullptr;
    switch (cls[0]) {
        case 1: kern = expr_kernel<1>(cls[1], cls[2], cls[3]); break;
        case 2: kern = expr_kernel<2>(cls[1], cls[2], cls[3]); break;
        case 3: kern = expr_kernel<4>(cls[1], cls[2], cls[3]); break;
        default: kern = expr_kernel<8>(cls[1], cls[2], cls[3]); break;
    }
    if (!kern) return set_error(EC_ERR_ARG, "%s: no kernel for stream classes %d %d %d %d", what, cls[0], cls[1], cls[2], cls[3]);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), static_cast<unsigned>(tuning().fused_lds_kb.load()) << 10, s, ea, out, out_mask, n);
    return check_launch("expr");
}

extern "C" ec_status ec_expr(const ec_dtype* dt, const void* const* p, int32_t n_streams, const ec_value* scalars, int32_t n_scalars,
                             const ec_expr_step* steps, int32_t n_steps, size_t n, double* out, ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (n == 0) return EC_OK;
    return launch_expr(dt, p, n_streams, nullptr, scalars, n_scalars, steps, n_steps, n, out, nullptr, static_cast<hipStream_t>(stream), "ec_expr");
}

extern "C" ec_status ec_masked_expr(const ec_dtype* dt, const void* const* p, const uint8_t* const* masks, int32_t n_streams,
                                    const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, size_t n,
                                    double* out, uint8_t* out_mask, ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (n == 0) return EC_OK;
    if (!masks) return set_error(EC_ERR_ARG, "ec_masked_expr: null masks");
    return launch_expr(dt, p, n_streams, masks, scalars, n_scalars, steps, n_steps, n, out, out_mask, static_cast<hipStream_t>(stream), "ec_masked_expr");
}

// BufferOps::min_max (src/buffer.rs:169-173; masked: src/masked/masked_buffer.rs:208-217) of a program's result without its
// raster.  Compiled form: the generated kernel folds order keys of the values it computes (ec_expr_jit.hip, reduce variant) —
// the streams are read, nothing is written but 16 bytes.  Until a program is compiled (and with expr_jit = 0): the program into
// a temporary from the pool, then ec_min_max of it — two passes, same answer.
// {~key(min), key(max)} of the program's result into DEVICE memory `keys2_dev`, asynchronously on `s` — the payload of the
// sharded reduction (one element-wise MAX all-reduce combines the shards), and what ec_expr_min_max waits for and decodes.
static const int64_t kF64Identities[2] = {~order_key<double>(std::numeric_limits<double>::max()), order_key<double>(std::numeric_limits<double>::lowest())};

static ec_status expr_min_max_keys(const ec_dtype* dt, const void* const* p, const uint8_t* const* masks_or_null, int32_t n_streams,
                                   const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, size_t n,
                                   int64_t* keys2_dev, hipStream_t s, const char* what) {
    ec_status st;
    {   // the program is checked even when there is nothing to fold
        ExprArgs probe{};
        if ((st = program_of(probe, dt, n_streams, n_scalars, steps, n_steps, what)) != EC_OK) return st;
    }
    if (!keys2_dev) return set_error(EC_ERR_ARG, "%s: null keys", what);
    // folded from (f64::MAX, f64::MIN): the identities first (a static source: the copy may still be in flight on return)
    if ((st = check_hip(hipMemcpyAsync(keys2_dev, kF64Identities, sizeof kF64Identities, hipMemcpyHostToDevice, s), "hipMemcpyAsync(keys)")) != EC_OK) return st;
    if (n == 0) return EC_OK;
    ExprArgs ea{};
    bool aligned = true;
    int cls[kExprMaxStreams] = {0, 0, 0, 0};
    if ((st = prepare_expr(ea, &aligned, cls, dt, p, n_streams, masks_or_null, scalars, n_scalars, steps, n_steps, n, what)) != EC_OK) return st;
    if (aligned) {
        bool compiled = false;
        if ((st = expr_jit_launch(ea, n, nullptr, nullptr, keys2_dev, s, &compiled)) != EC_OK) return st;
        if (compiled) return EC_OK;
    }
    // two passes: the program into a temporary from the pool, ec_min_max_keys of it (which overwrites the identities), the
    // temporary returned in stream order
    void *tmp = nullptr, *tmask = nullptr;
    if ((st = ec_alloc_async(&tmp, n * sizeof(double), s)) != EC_OK) return st;
    if (masks_or_null && (st = ec_alloc_async(&tmask, n, s)) != EC_OK) {
        (void)ec_free_async(tmp, s);
        return st;
    }
    st = launch_expr(dt, p, n_streams, masks_or_null, scalars, n_scalars, steps, n_steps, n, static_cast<double*>(tmp), static_cast<uint8_t*>(tmask), s, what);
    if (st == EC_OK) st = ec_min_max_keys(EC_F64, tmp, static_cast<const uint8_t*>(tmask), n, keys2_dev, s);
    const std::string keep = st != EC_OK ? last_error_text() : std::string();
    (void)ec_free_async(tmp, s);
    if (tmask) (void)ec_free_async(tmask, s);
    return st != EC_OK ? set_error_text(st, keep) : EC_OK;
}

extern "C" ec_status ec_expr_min_max_keys(const ec_dtype* dt, const void* const* p, const uint8_t* const* masks_or_null, int32_t n_streams,
                                          const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, size_t n,
                                          int64_t* keys2_dev, ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    return expr_min_max_keys(dt, p, masks_or_null, n_streams, scalars, n_scalars, steps, n_steps, n, keys2_dev, static_cast<hipStream_t>(stream),
                             "ec_expr_min_max_keys");
}

// BufferOps::min_max (src/buffer.rs:169-173; masked: src/masked/masked_buffer.rs:208-217) of a program's result without its
// raster.  Compiled form: the generated kernel folds order keys of the values it computes (ec_expr_jit.hip, reduce variant) —
// the streams are read, nothing is written but 16 bytes.  Until a program is compiled (and with expr_jit = 0): two passes.
extern "C" ec_status ec_expr_min_max(const ec_dtype* dt, const void* const* p, const uint8_t* const* masks_or_null, int32_t n_streams,
                                     const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, size_t n,
                                     ec_value* mn, ec_value* mx, ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (!mn || !mx) return set_error(EC_ERR_ARG, "ec_expr_min_max: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the two key words live in the stream's reduction scratch, after the words ec_min_max_keys itself uses.  (Not a 16-byte
    // block from the pool: hipMallocAsync carves such a block out of the 2 GiB one the two-pass form has just returned, and the
    // next 2 GiB request becomes a fresh 130 ms allocation — profiles/r03/expr_kernel.md.)
    Scratch sc;
    if ((st = get_scratch(s, &sc)) != EC_OK) return st;
    int64_t keys[2];
    {
        std::lock_guard<std::mutex> turn(*sc.mu);
        int64_t* dkeys = sc.dev_result() + 2;
        st = expr_min_max_keys(dt, p, masks_or_null, n_streams, scalars, n_scalars, steps, n_steps, n, dkeys, s, "ec_expr_min_max");
        if (st == EC_OK) st = check_hip(hipMemcpyAsync(keys, dkeys, sizeof keys, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(keys)");
        if (st == EC_OK) st = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");  // `keys` is this frame's: nothing may be in flight on return
    }
    if (st != EC_OK) return st;
    return ec_min_max_decode(EC_F64, keys, mn, mx);
}

// Diagnostics, no device needed: the source the library would compile for this program (all streams non-temporal), and a
// trial compile of it when `arch_or_null` names a processor.
extern "C" ec_status ec_expr_source(const ec_dtype* dt, int32_t n_streams, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps,
                                    const char* arch_or_null, char* buf, size_t cap, size_t* len) {
    ExprArgs ea{};
    ec_status st = program_of(ea, dt, n_streams, n_scalars, steps, n_steps, "ec_expr_source");
    if (st != EC_OK) return st;
    const char* variant = std::getenv("EC_EXPR_SOURCE_VARIANT");  // "reduce": the min_max variant (ec_expr_min_max) instead
    std::string src;
    {   // what the library makes of the program before it compiles anything: its canonical tree, and the built-in straight-line
        // kernel that serves it if the tree is in the ahead-of-time catalogue (ec_expr_fixed.hpp) and the streams share one width
        FixedMap fm;
        int id = -1;
        const std::string tree = expr_fixed_tree(ea, &fm, &id);
        static const char* const kName[kFixCount] = {"NDVI", "add-mul", "EVI", "affine"};
        bool one_width = id >= 0;
        for (int k = 1; one_width && k < n_streams; ++k) one_width = ecl::size_of(dt[k]) == ecl::size_of(dt[0]);
        src = "// tree: " + (tree.empty() ? std::string("(none within the catalogue's size)") : tree) + "\n// ahead-of-time kernel: " +
              (id < 0 ? std::string("none (not in the catalogue)") : one_width ? std::string(kName[id]) : std::string(kName[id]) + " in the catalogue, but the streams differ in width: none") + "\n";
    }
    src += expr_jit_source(ea, variant && !std::strcmp(variant, "reduce"));
    if (len) *len = src.size() + 1;
    if (buf && cap > 0) {
        const size_t k = src.size() < cap - 1 ? src.size() : cap - 1;
        std::memcpy(buf, src.data(), k);
        buf[k] = 0;
    }
    if (arch_or_null) {
        std::string code, log;
        return expr_jit_compile(src, arch_or_null, &code, &log);  // the two comment lines in front compile to nothing
    }
    return EC_OK;
}

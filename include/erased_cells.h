/*
 * erased_cells.h — C ABI of liberased_cells_hip.so: the MI355X (gfx950) drop-in
 * for the per-cell arithmetic path of the Rust crate `erased-cells` 0.1.1.
 *
 * The reference has no FFI seam of its own; the seam is its operator/trait
 * surface.  Each entry point below replaces the BODY of one reference function
 * (cited as path:line relative to the reference tree); the host language keeps
 * the dtype-erased dispatch (`with_ct!`, src/lib.rs:85-101) and the shape rules
 * (zip truncation, empty -> UInt8, length asserts) and calls in here with plain
 * device pointers and sizes.  INTEGRATION.md shows the Rust `extern "C"` block
 * and the operator impls a maintainer would add.
 *
 * Conventions
 *  - Every `const void*` / `void*` data argument is a DEVICE pointer (ec_alloc,
 *    hipMalloc or any other HIP allocation) unless the name says `host`.  It needs
 *    only its cell type's natural alignment: windows at any cell offset are fine.
 *  - Inputs are never written.  Outputs are caller-allocated; the library keeps
 *    no hidden buffers except a small per-stream scratch for reduction partials.
 *  - All compute entry points are asynchronous on `stream` (a hipStream_t; NULL
 *    = the default stream) except those documented as "synchronous result".
 *  - Every function returns ec_status; EC_OK == 0.  No exceptions cross the ABI.
 *    ec_last_error_string() gives the thread-local detail of the last failure.
 *  - dtype codes are `CellType as u8` (src/ctype.rs:11-20, order of
 *    src/lib.rs:89-98): UInt8=0 .. Float64=9.
 *  - Masks are `Vec<bool>` images: one byte per cell, 0 or 1 (src/masked/mask.rs:10-12).
 */
#ifndef ERASED_CELLS_H
#define ERASED_CELLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EC_ABI_VERSION 1

typedef int32_t ec_status;
enum {
    EC_OK = 0,
    EC_ERR_NARROWING = 1,        /* Error::NarrowingError{src,dst}      src/error.rs:14-15 */
    EC_ERR_UNSUPPORTED_TYPE = 2, /* Error::UnsupportedCellTypeError     src/error.rs:16-17 */
    EC_ERR_LENGTH = 3,           /* the reference panics (assert_eq! masked_buffer.rs:48-53) */
    EC_ERR_HIP = 4,              /* a HIP runtime call failed */
    EC_ERR_RCCL = 5,             /* reserved for the collective layer */
    EC_ERR_ARG = 6,              /* null pointer / bad enum */
    EC_ERR_NOT_INITIALIZED = 7
};

/* CellType (src/ctype.rs:11-20). */
typedef uint8_t ec_dtype;
enum {
    EC_U8 = 0, EC_U16 = 1, EC_U32 = 2, EC_U64 = 3,
    EC_I8 = 4, EC_I16 = 5, EC_I32 = 6, EC_I64 = 7,
    EC_F32 = 8, EC_F64 = 9, EC_NTYPES = 10
};

/* Add/Sub/Mul/Div of cb_bin_op! (src/buffer.rs:355-358). */
typedef int32_t ec_op;
enum { EC_ADD = 0, EC_SUB = 1, EC_MUL = 2, EC_DIV = 3 };

/* CellValue (src/value.rs:12-20): tag + payload, 16 bytes. */
typedef struct ec_value {
    uint8_t dtype;
    uint8_t pad_[7];
    union {
        uint8_t u8; uint16_t u16; uint32_t u32; uint64_t u64;
        int8_t i8; int16_t i16; int32_t i32; int64_t i64;
        float f32; double f64;
        uint64_t bits;
    } v;
} ec_value;

typedef void *ec_stream; /* hipStream_t */

/* ---------------------------------------------------------------- *
 * Type lattice — pure host functions, no device needed.
 * ---------------------------------------------------------------- */
ec_dtype ec_union(ec_dtype a, ec_dtype b);          /* CellType::union        src/ctype.rs:99-126  */
int32_t ec_can_fit_into(ec_dtype src, ec_dtype dst); /* CellType::can_fit_into src/ctype.rs:129-131 */
size_t ec_size_of(ec_dtype t);                      /* CellType::size_of      src/ctype.rs:87-96   */
ec_dtype ec_neg_result_type(ec_dtype t);            /* result variant of Neg  src/value.rs:224-240 */
ec_status ec_min_value(ec_dtype t, ec_value *out);  /* CellType::min_value    src/ctype.rs:158-167 */
ec_status ec_max_value(ec_dtype t, ec_value *out);  /* CellType::max_value    src/ctype.rs:170-179 */
/* NoData::<T>::Default.value() (src/masked/nodata.rs:27-38): T::MIN / canonical NaN. */
ec_status ec_nodata_default(ec_dtype t, ec_value *out);
/* CellValue::convert (src/value.rs:74-98) for scalars (host). */
ec_status ec_value_convert(const ec_value *v, ec_dtype dst, ec_value *out);
/* CellValue::to_f64 (src/value.rs:145-156). */
double ec_value_to_f64(const ec_value *v);

/* Row-block sharding: rows [g*R/G, (g+1)*R/G) with the first
 * R mod G shards one row longer; returns the shard's [cell_offset, cell_len). */
ec_status ec_shard_range(uint64_t n_rows, uint64_t n_cols, uint32_t shard, uint32_t n_shards,
                         uint64_t *cell_offset, uint64_t *cell_len);

/* ---------------------------------------------------------------- *
 * Runtime: device, memory, streams, errors.
 * ---------------------------------------------------------------- */
int32_t ec_abi_version(void);
/* The runtime is per device.  ec_init(device) sets up `device` (its CU count, a stream-ordered memory pool the
 * library owns, per-stream reduction scratch) and makes it the calling thread's current library device; it may be
 * called for several devices (one process driving all GPUs of a node) and is idempotent.  Host threads that never
 * chose a device use the first one the process initialised — the one-process-per-GPU shape needs nothing else.
 * ec_set_device switches the calling thread between initialised devices: allocations, streams and launches of a
 * thread go to its current library device.  Device pointers and streams must be used with their own device. */
ec_status ec_init(int32_t device);
ec_status ec_set_device(int32_t device);
ec_status ec_get_device(int32_t *device);
ec_status ec_shutdown(void);         /* waits for every initialised device, frees scratch, trims and destroys the pools */
const char *ec_last_error_string(void);
/* src/dst of the last EC_ERR_NARROWING on this thread (Error::NarrowingError fields). */
ec_status ec_last_narrowing(ec_dtype *src, ec_dtype *dst);
ec_status ec_device_info(int32_t *n_cu, uint64_t *hbm_bytes, char *name, size_t name_cap);

ec_status ec_alloc(void **dptr, size_t bytes);
ec_status ec_free(void *dptr);
/* Stream-ordered allocation from the library's own pool on the current device (hipMallocFromPoolAsync /
 * hipFreeAsync; the device's default pool is never touched): what the host
 * mirrors use for operator results, which the reference allocates per call (`collect()`,
 * src/buffer.rs:327).  A block may be used by work enqueued on `stream` after the call, and freed
 * blocks are recycled without synchronising the device. */
ec_status ec_alloc_async(void **dptr, size_t bytes, ec_stream stream);
/* `stream` must be ordered after the last use of the block (the stream that used it last, normally). */
ec_status ec_free_async(void *dptr, ec_stream stream);
/* Free of a block allocated on `alloc_stream` whose last use was enqueued on `last_use_stream`: the free is queued on
 * `alloc_stream` after an event recorded on `last_use_stream` — what a host mirror's destructor calls, which cannot
 * know on which stream the caller works by the time a buffer is dropped. */
ec_status ec_free_ordered(void *dptr, ec_stream alloc_stream, ec_stream last_use_stream);
/* The pool caches freed blocks up to its release threshold (ec_tune_set("pool_keep_mb"), default 32768 per
 * device); ec_pool_trim waits for the current device and returns everything above `keep_bytes` to the driver. */
ec_status ec_pool_trim(size_t keep_bytes);
ec_status ec_upload(void *dst_dev, const void *src_host, size_t bytes, ec_stream stream);   /* From<Vec<T>> */
ec_status ec_download(void *dst_host, const void *src_dev, size_t bytes, ec_stream stream); /* to_vec; waits for completion */
ec_status ec_copy(void *dst_dev, const void *src_dev, size_t bytes, ec_stream stream);      /* Clone (buffer.rs:151-153) */
ec_status ec_stream_create(ec_stream *out);
/* Allocates the per-stream reduction scratch of a stream the library did not create (NULL = default
 * stream, a torch stream, ...).  After this, every asynchronous entry point — including ec_min_max_keys and
 * ec_mask_counts_device — performs no allocation and no synchronisation (ec_fused over combinations of cell types
 * without a one-pass kernel excepted: it draws temporaries from the stream-ordered pool), so a chain of calls can be captured into a
 * hipGraph on that stream and replayed. */
ec_status ec_prepare_stream(ec_stream stream);
/* Releases that scratch again (the library holds scratch for at most 64 streams per device and recycles the
 * least recently used entry beyond that, so calling this for a dying foreign stream is optional). */
ec_status ec_release_stream(ec_stream stream);
ec_status ec_stream_destroy(ec_stream s);
ec_status ec_stream_sync(ec_stream s);

/* ---------------------------------------------------------------- *
 * Element-wise arithmetic.  out[i] = f64(l[i]) op f64(r[i]), always Float64
 * (src/value.rs:199-217).  `n` = min(len l, len r): the caller applies the zip
 * truncation of src/buffer.rs:327.  n == 0 is a no-op.
 * ---------------------------------------------------------------- */
/* impl {Add,Sub,Mul,Div} for &CellBuffer — src/buffer.rs:324-329 */
ec_status ec_binop(ec_op op, ec_dtype lt, const void *l, ec_dtype rt, const void *r,
                   size_t n, double *out, ec_stream stream);
/* impl $trt<R: Into<CellValue>> for CellBuffer — src/buffer.rs:346-352 */
ec_status ec_binop_scalar(ec_op op, ec_dtype lt, const void *l, size_t n, const ec_value *rhs,
                          double *out, ec_stream stream);
/* impl $trt for &MaskedCellBuffer — src/masked/masked_buffer.rs:326-335: the
 * buffer op over ALL cells and `lmask & rmask` (mask.rs:129-140) in one launch. */
ec_status ec_masked_binop(ec_op op, ec_dtype lt, const void *l, const uint8_t *lmask,
                          ec_dtype rt, const void *r, const uint8_t *rmask, size_t n,
                          double *out, uint8_t *out_mask, ec_stream stream);
/* impl Neg for &CellBuffer — src/buffer.rs:360-365 + src/value.rs:224-240.
 * `out` holds n cells of ec_neg_result_type(t). Signed MIN wraps (release build). */
ec_status ec_neg(ec_dtype t, const void *in, size_t n, void *out, ec_stream stream);
/* BufferOps::convert — src/buffer.rs:150-167. EC_ERR_NARROWING iff
 * !can_fit_into(st, dt), decided before any device work; st == dt is a copy. */
ec_status ec_convert(ec_dtype st, const void *src, ec_dtype dt, void *dst, size_t n, ec_stream stream);
/* BufferOps::fill — src/buffer.rs:79-88 (value->dtype must equal t). */
ec_status ec_fill(ec_dtype t, void *dst, size_t n, const ec_value *value, ec_stream stream);

/* Fused two-level expression:  out[i] = (x[i] o1 y[i]) o2 (z[i] o3 w[i])  in one pass,
 * operands dt[k]/p[k] for k = x, y, z, w.  Each step is the same correctly rounded f64 op the eager
 * chain of impl $trt for &CellBuffer (src/buffer.rs:324-329) performs, so the result is bit-identical
 * to evaluating the three operators one by one (e.g. NDVI `(&nir - &red) / (nir + red)`,
 * src/gdal/rasterband.rs:148) while the f64 temporaries never reach HBM.  o3 == EC_OP_NONE: the second
 * term is z alone (`(x o1 y) o2 z`, e.g. `(a + b) * c`) and dt[3]/p[3] are ignored.  Operands may alias.
 * An operand with p[k] == NULL is the scalar `scalars[k]` (the RHS-scalar form of src/buffer.rs:346-352,
 * e.g. `(buf + ones) * 2.0`, examples/masked.rs:12); at least one operand must be a buffer; `n` is the
 * shortest buffer operand's length.  Every call is ONE pass over its operands with no allocation and no
 * synchronisation — whatever the mix of cell types, aliases and scalars — so every call can be captured in a
 * hipGraph: operands of one cell type run a kernel specialised for that type and op triple, every other mix a
 * kernel specialised for the operands' byte widths that widens each cell to f64 in registers (the reference's
 * `unify` + `to_f64`, src/value.rs:103-107,207).  (ec_tune_set("fused_mixed", 0) selects the comparison path of
 * rounds 1-2 instead: convert mixed operands to their CellType::union into pooled temporaries, then fuse.) */
#define EC_OP_NONE (-1)
ec_status ec_fused(ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void *const p[4],
                   const ec_value *scalars_or_null, size_t n, double *out, ec_stream stream);
/* The masked form: also out_mask = AND of the buffer operands' masks, as the eager chain of
 * impl $trt for &MaskedCellBuffer (src/masked/masked_buffer.rs:326-364) produces (a scalar has no mask). */
ec_status ec_masked_fused(ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void *const p[4],
                          const uint8_t *const masks[4], const ec_value *scalars_or_null, size_t n,
                          double *out, uint8_t *out_mask, ec_stream stream);

/* Expression programs (beyond two levels): a whole operator tree over up to four buffers in ONE pass.
 * The reference evaluates `2.5 * (nir - red) / (nir + 6.0 * red - 7.5 * blue + 1.0)` operator by operator, a pass and an
 * f64 temporary each (impl $trt for &CellBuffer / for CellBuffer with a scalar, src/buffer.rs:324-352); here the same
 * operators run in the same order on registers, every step the same correctly rounded f64 op, so the result is
 * bit-identical to the eager evaluation.  A program is a list of steps `reg[dst] = a op b` over four registers; an
 * operand refers to a stream (buffer k of the call), a register an earlier step wrote, or a scalar; the program's value
 * is what its LAST step computed.  `n` is the shortest buffer's length (zip truncation at every step).  Streams may have
 * any mix of cell types; no allocation, no synchronisation (graph-capturable).  EC_ERR_ARG for a malformed program
 * (bad reference, a register read before it is written, counts out of range). */
typedef struct ec_expr_step {
    int8_t op;  /* EC_ADD .. EC_DIV */
    int8_t a;   /* left operand:  EC_EXPR_STREAM(k) | EC_EXPR_REG(k) | EC_EXPR_SCALAR(k) */
    int8_t b;   /* right operand */
    int8_t dst; /* register 0..3 that receives the result */
} ec_expr_step;
#define EC_EXPR_STREAM(k) ((int8_t)(k))       /* k = 0..3: dt[k] / p[k] of the call */
#define EC_EXPR_REG(k) ((int8_t)(4 + (k)))    /* k = 0..3 */
#define EC_EXPR_SCALAR(k) ((int8_t)(8 + (k))) /* k = 0..7: scalars[k] */
enum { EC_EXPR_MAX_STREAMS = 4, EC_EXPR_REGS = 4, EC_EXPR_MAX_SCALARS = 8, EC_EXPR_MAX_STEPS = 16 };
ec_status ec_expr(const ec_dtype *dt, const void *const *p, int32_t n_streams, const ec_value *scalars,
                  int32_t n_scalars, const ec_expr_step *steps, int32_t n_steps, size_t n, double *out,
                  ec_stream stream);
/* The masked form: also out_mask = AND of the streams' masks, what the eager chain of impl $trt for &MaskedCellBuffer
 * (src/masked/masked_buffer.rs:326-364) leaves behind (scalars carry no mask). */
ec_status ec_masked_expr(const ec_dtype *dt, const void *const *p, const uint8_t *const *masks, int32_t n_streams,
                         const ec_value *scalars, int32_t n_scalars, const ec_expr_step *steps, int32_t n_steps,
                         size_t n, double *out, uint8_t *out_mask, ec_stream stream);
/* BufferOps::min_max (src/buffer.rs:169-173; masked: src/masked/masked_buffer.rs:208-217) of a program's RESULT without
 * its raster: (min, max) under total_cmp of the f64 cells the program computes, over the cells that are valid under the AND
 * of masks_or_null (NULL: all cells), folded from (f64::MAX, f64::MIN) like every min_max here.  Once the program is
 * compiled for itself (below) the streams are read and nothing is written — NDVI's statistics cost the bands' 4 B/cell
 * instead of 4 + 8 + 8; until then (and with expr_jit = 0) the library runs the program into a temporary from its pool and
 * reduces that: two passes, the same answer.  Synchronous. */
ec_status ec_expr_min_max(const ec_dtype *dt, const void *const *p, const uint8_t *const *masks_or_null, int32_t n_streams,
                          const ec_value *scalars, int32_t n_scalars, const ec_expr_step *steps, int32_t n_steps,
                          size_t n, ec_value *mn, ec_value *mx, ec_stream stream);
/* Asynchronous: {~key(min), key(max)} of the same into DEVICE memory keys2_dev, as ec_min_max_keys writes them for a
 * buffer — the 16-byte payload one element-wise MAX all-reduce combines across shards; decode with
 * ec_min_max_decode(EC_F64, ...). */
ec_status ec_expr_min_max_keys(const ec_dtype *dt, const void *const *p, const uint8_t *const *masks_or_null,
                               int32_t n_streams, const ec_value *scalars, int32_t n_scalars, const ec_expr_step *steps,
                               int32_t n_steps, size_t n, int64_t *keys2_dev, ec_stream stream);

/* How ec_expr runs a program.  Three forms, the same cells from each.  (1) Ahead of time: the programs whose TREE is one
 * of NDVI `(a - b) / (a + b)`, `(a + b) * c`, EVI `((a - b) * k0) / (((a + b * k1) - c * k2) + k3)` and `a * k0 + k1`, over
 * distinct streams of one cell width, are built into the library as straight-line kernels (csrc/ec_expr_fixed.hpp) and run
 * that way from their first launch, inside stream captures and without hiprtc; register names, the schedule of independent
 * sub-trees, the numbering of streams and scalars and the side a lone scalar of + or * stands on do not matter
 * (ec_tune_set("expr_fixed", 0) turns this off; ec_stat_get "expr_fixed_launches").  (2) The kernel that serves every
 * other program is an interpreter (a step is decoded once per wave):
 * bound by instruction issue, ≈ 0.04 ms per step over 16384^2 cells beyond the first.  (3) A program is launch-uniform, so
 * the library can also compile it for itself — straight-line code with typed loads, through hiprtc (resolved lazily;
 * without it the interpreter keeps serving) — and cache the module per (program, cell types, load policy).  Same cells
 * either way.  ec_tune_set("expr_jit", v): 0 = never compile; 1 (default) = compile on a background thread once a
 * program has interpreted 2^31 cell-steps, launches interpret until the module is ready; 2 = compile on the calling
 * thread at first sight (≈ 0.3 s per program, 3 s for the first) and fail loudly if that fails.  The environment variable
 * EC_HIPRTC_LIB names the hiprtc library to load instead of libhiprtc.so.  Inside a stream
 * capture a program whose module is not loaded yet is interpreted.  ec_stat_get: "expr_interp_launches",
 * "expr_jit_launches", "expr_jit_compiles", "expr_jit_failures", "expr_jit_programs".
 *
 * ec_expr_source (diagnostics; needs no device): the HIP source the library would compile for the program (every stream
 * non-temporal) into buf[0..cap), *len = bytes needed with the terminating 0; with `arch_or_null` (e.g. "gfx950") also
 * a trial compile, EC_ERR_HIP and the compiler's log in ec_last_error_string() if it fails.  (With the environment
 * variable EC_EXPR_SOURCE_VARIANT=reduce: the variant ec_expr_min_max compiles.) */
ec_status ec_expr_source(const ec_dtype *dt, int32_t n_streams, int32_t n_scalars, const ec_expr_step *steps,
                         int32_t n_steps, const char *arch_or_null, char *buf, size_t cap, size_t *len);

/* Host memory in, host memory out.  The reference's operands and results are Vecs in host memory
 * (src/buffer.rs:12-55, impl $trt for &CellBuffer :324-352); a caller that keeps nothing resident is bound by PCIe —
 * the operands' bytes up, 8 bytes per cell down — not by the kernel.  ec_host_expr evaluates an expression program
 * (ec_expr: a single operator is a one-step program) over HOST arrays p_host[k] of n cells into out_host[0..n): chunks
 * of `chunk_cells` cells (0 = 2^25), upload / kernel / download on three streams over double-buffered device staging
 * (two slots of chunk_cells x (the operands' bytes + 8 [+ masks]) from the library's pool: 2 x 370 MB for u8 / u16 at 2^25),
 * so both directions of the link are busy at once.  Page-locked buffers (ec_host_alloc) are copied asynchronously as
 * they are; any other buffer is page-locked for the duration of the call (hipHostRegister: about one pass over the
 * pages) and, if that is refused, copied through the runtime's pageable path.  Operands may be windows of one array and
 * may overlap each other; out_host may be one of the f64 operands itself (same address), not a shifted window of one.
 * A call that moves at most 64 MiB (and leaves chunk_cells 0) is not worth a pipeline (≈ 1 ms of set-up): it runs as one
 * chunk on the calling thread's own stream (hipStreamPerThread).
 * Synchronous; uses its own streams; may be called from several host threads at once. */
ec_status ec_host_alloc(void **hptr, size_t bytes); /* page-locked host memory */
ec_status ec_host_free(void *hptr);
ec_status ec_host_expr(const ec_dtype *dt, const void *const *p_host, int32_t n_streams, const ec_value *scalars,
                       int32_t n_scalars, const ec_expr_step *steps, int32_t n_steps, size_t n, double *out_host,
                       size_t chunk_cells);
/* The masked form, host to host: MaskedCellBuffer::from_vec_with_nodata (src/masked/masked_buffer.rs:62-71) of every
 * stream — nodata[k] typed as dt[k], or NULL for NoData::None — the program with the AND of the masks (:326-364), and
 * to_vec_with_nodata (:137-152) of the f64 result: out_host[i] = valid ? value : *out_nodata_or_null (NULL: the values of
 * all cells, as the reference computes them); out_mask_host_or_null receives the result's mask (1 = valid).  The masks
 * are derived and applied on the device: what crosses the link is still the operands up and 8 (+ 1) bytes per cell down. */
ec_status ec_host_masked_expr(const ec_dtype *dt, const void *const *p_host, const ec_value *const *nodata,
                              int32_t n_streams, const ec_value *scalars, int32_t n_scalars,
                              const ec_expr_step *steps, int32_t n_steps, size_t n, double *out_host,
                              const double *out_nodata_or_null, uint8_t *out_mask_host_or_null, size_t chunk_cells);

/* ---------------------------------------------------------------- *
 * min/max under the reference's total order (ints natural; floats total_cmp),
 * folded from (T::MAX, T::MIN) — src/buffer.rs:169-173, masked:
 * src/masked/masked_buffer.rs:208-217 (mask may be NULL).
 * ---------------------------------------------------------------- */
/* Synchronous result: waits for the stream, returns host values typed `t`. */
ec_status ec_min_max(ec_dtype t, const void *p, const uint8_t *mask_or_null, size_t n,
                     ec_value *mn, ec_value *mx, ec_stream stream);
/* Asynchronous: writes two order-preserving int64 keys {~key(min), key(max)}
 * to DEVICE memory `keys2_dev`, so that an element-wise MAX all-reduce across
 * shards (RCCL over xGMI) followed by ec_min_max_decode gives the global answer. */
ec_status ec_min_max_keys(ec_dtype t, const void *p, const uint8_t *mask_or_null, size_t n,
                          int64_t *keys2_dev, ec_stream stream);
ec_status ec_min_max_decode(ec_dtype t, const int64_t keys2_host[2], ec_value *mn, ec_value *mx);

/* ---------------------------------------------------------------- *
 * The sharded path: a raster is cut into contiguous row-blocks (ec_shard_range), one per GPU;
 * element-wise kernels are local to a shard, and the only exchange is the all-reduce over xGMI of the
 * 16-byte reduction payloads written by ec_min_max_keys (MAX over two int64) and ec_mask_counts_device
 * (SUM over two uint64).  The reference has no counterpart (it is single-threaded CPU code); what is
 * reduced is BufferOps::min_max (src/buffer.rs:169-173) and Mask::counts (src/masked/mask.rs:72-80).
 * ---------------------------------------------------------------- */
typedef void *ec_comm; /* ncclComm_t; librccl is loaded lazily, EC_ERR_RCCL if it is missing or a call fails */
typedef struct ec_comm_uid { char bytes[128]; } ec_comm_uid; /* ncclUniqueId */
/* One process per GPU: rank 0 makes the id, the host program hands its 128 bytes to the other ranks (a file, a
 * socket, MPI, ...), every rank calls ec_comm_init_rank with its current library device bound (blocks until all
 * n_ranks have joined). */
ec_status ec_comm_get_unique_id(ec_comm_uid *uid);
ec_status ec_comm_init_rank(const ec_comm_uid *uid, int32_t n_ranks, int32_t rank, ec_comm *comm);
/* One process, n GPUs: ec_init()s every listed device and builds the clique (ncclCommInitAll). */
ec_status ec_comm_init_all(const int32_t *devices, int32_t n, ec_comm *comms);
ec_status ec_comm_destroy(ec_comm comm);
/* In-place all-reduce of the device payloads; asynchronous on `stream` (the stream the payload was produced on).
 * (bench.py and the Python mirror use torch.distributed's all_reduce, which is the same RCCL call.) */
ec_status ec_allreduce_min_max_keys(ec_comm comm, int64_t *keys2_dev, ec_stream stream);
ec_status ec_allreduce_counts(ec_comm comm, uint64_t *counts2_dev, ec_stream stream);

/* A shard group: one process driving n GPUs — per device a launch thread bound to it, a stream, a 32-byte payload
 * slot and (unless EC_GROUP_HOST_COMBINE) an RCCL communicator of the n-device clique.  Shard i of every sharded
 * call lives on device i of the group.
 *   Element-wise calls (ec_sharded_binop, _masked_binop, _convert, _mask_from_nodata, _fused) are fire-and-forget:
 * the arguments are checked on the calling thread (EC_ERR_ARG / EC_ERR_UNSUPPORTED_TYPE / EC_ERR_NARROWING come back at
 * once), the per-shard pointer arrays are copied, one job per device is queued for its launch thread, and the call
 * returns without waiting for the launches to be issued: the host can queue the next call while the threads issue
 * this one side by side (host cost per call and per device: profiles/r03/group_fanout.md).  Calls on one group are
 * issued on every device in the order they were made.  A failure inside a queued job (a launch error) is kept by
 * the group and returned — once — by the next ec_shard_group_sync, ec_sharded_min_max or ec_sharded_counts.
 * The device memory named in a call must stay allocated until the group has been synchronised or it is released
 * through ec_sharded_free (which queues behind the launches).  EC_GROUP_BLOCKING_ISSUE restores the form of rounds
 * 1-2: every call waits until all launch threads have issued and returns their first failing status itself.
 *   The reductions return a value and therefore wait; they run in phases (check every shard's arguments, reduce
 * locally on every shard, and only when all of that succeeded exchange the payloads), so a shard that fails cannot
 * leave the others waiting inside a collective.  Should a communicator fail between the ranks' enqueues, the group
 * is poisoned (communicators aborted, every later call EC_ERR_RCCL) rather than left to hang.
 *   EC_GROUP_HOST_COMBINE folds the n 16-byte payloads on the host instead of over xGMI (no RCCL needed; also the
 * only way to list one device twice, e.g. to rehearse on a 1-GPU box). */
typedef struct ec_shard_group ec_shard_group;
enum { EC_GROUP_RCCL = 0, EC_GROUP_HOST_COMBINE = 1, EC_GROUP_BLOCKING_ISSUE = 2 };
ec_status ec_shard_group_create(const int32_t *devices, int32_t n, uint32_t flags, ec_shard_group **out);
ec_status ec_shard_group_destroy(ec_shard_group *g);
int32_t ec_shard_group_size(const ec_shard_group *g);
ec_status ec_shard_group_shard(const ec_shard_group *g, int32_t shard, int32_t *device, ec_stream *stream);
/* Runs fn(shard, device, stream, user) once per shard, each on that shard's launch thread with its device current,
 * concurrently; returns when all have returned (the work they enqueued may still be running): the building block
 * with which a host fans out any entry point of this header.  The first failing status is returned. */
/* (fn must not call ec_shard_group_* / ec_sharded_* of the SAME group — one sharded call runs at a time per group;
 * destroy every group before ec_shutdown().) */
typedef ec_status (*ec_shard_fn)(int32_t shard, int32_t device, ec_stream stream, void *user);
ec_status ec_shard_group_foreach(ec_shard_group *g, ec_shard_fn fn, void *user);
/* Waits until everything queued so far has been issued and every shard's stream has drained; returns (and clears)
 * the first failure a fire-and-forget call left behind. */
ec_status ec_shard_group_sync(ec_shard_group *g);
/* Counters of a group: "jobs_posted" (fire-and-forget jobs queued so far), "poisoned", "blocking_issue". */
ec_status ec_shard_group_stat(const ec_shard_group *g, const char *key, int64_t *value);
/* Per-shard device blocks (bytes[i] on device i) and the scatter / gather of one host buffer by byte ranges
 * (`From<Vec<T>>` / `to_vec` of a sharded buffer; the ranges come from ec_shard_range x sizeof(T)). */
ec_status ec_sharded_alloc(ec_shard_group *g, const size_t *bytes, void **dptrs);
ec_status ec_sharded_free(ec_shard_group *g, void *const *dptrs);
ec_status ec_sharded_upload(ec_shard_group *g, void *const *dst_dev, const void *src_host,
                            const size_t *byte_offsets, const size_t *bytes);
ec_status ec_sharded_download(ec_shard_group *g, void *dst_host, const void *const *src_dev,
                              const size_t *byte_offsets, const size_t *bytes);
/* impl {Add,Sub,Mul,Div} for &CellBuffer (src/buffer.rs:324-329) on every shard; fire-and-forget, no communication. */
ec_status ec_sharded_binop(ec_shard_group *g, ec_op op, ec_dtype lt, const void *const *l, ec_dtype rt,
                           const void *const *r, const size_t *n, double *const *out);
/* The other element-wise entry points on every shard, same arguments as their one-GPU forms with one pointer per
 * shard: impl $trt for &MaskedCellBuffer (src/masked/masked_buffer.rs:326-335), BufferOps::convert
 * (src/buffer.rs:150-167; EC_ERR_NARROWING before any device work), from_vec_with_nodata
 * (src/masked/masked_buffer.rs:62-71), and the fused chains — p[k] is operand k's per-shard pointer array, or NULL
 * for a scalar operand (scalars[k]); masks_or_null / out_mask_or_null both given or both NULL.  Fire-and-forget. */
ec_status ec_sharded_masked_binop(ec_shard_group *g, ec_op op, ec_dtype lt, const void *const *l,
                                  const uint8_t *const *lmask, ec_dtype rt, const void *const *r,
                                  const uint8_t *const *rmask, const size_t *n, double *const *out,
                                  uint8_t *const *out_mask);
ec_status ec_sharded_convert(ec_shard_group *g, ec_dtype st, const void *const *src, ec_dtype dt, void *const *dst,
                             const size_t *n);
ec_status ec_sharded_mask_from_nodata(ec_shard_group *g, ec_dtype t, const void *const *p, const size_t *n,
                                      const ec_value *nd_or_null, uint8_t *const *mask);
ec_status ec_sharded_fused(ec_shard_group *g, ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4],
                           const void *const *const p[4], const uint8_t *const *const masks_or_null[4],
                           const ec_value *scalars_or_null, const size_t *n, double *const *out,
                           uint8_t *const *out_mask_or_null);
/* Expression programs (ec_expr / ec_masked_expr) on every shard: p[k] (and masks_or_null[k]) is stream k's per-shard
 * pointer array; masks_or_null / out_mask_or_null both given or both NULL.  A malformed program is EC_ERR_ARG on the
 * calling thread, before anything is posted.  Fire-and-forget. */
ec_status ec_sharded_expr(ec_shard_group *g, const ec_dtype *dt, const void *const *const *p,
                          const uint8_t *const *const *masks_or_null, int32_t n_streams, const ec_value *scalars,
                          int32_t n_scalars, const ec_expr_step *steps, int32_t n_steps, const size_t *n,
                          double *const *out, uint8_t *const *out_mask_or_null);
/* Host memory in, host memory out over all the GPUs of the group: the row-blocks of an n_rows x n_cols raster
 * (ec_shard_range) go through ec_host_expr — or, with nodata_or_null, ec_host_masked_expr — side by side, one pipeline per
 * device: a host-resident raster is then bound by the sum of the GPUs' PCIe links, not by one.  Synchronous. */
ec_status ec_sharded_host_expr(ec_shard_group *g, const ec_dtype *dt, const void *const *p_host,
                               const ec_value *const *nodata_or_null, int32_t n_streams, const ec_value *scalars,
                               int32_t n_scalars, const ec_expr_step *steps, int32_t n_steps, uint64_t n_rows,
                               uint64_t n_cols, double *out_host, const double *out_nodata_or_null,
                               uint8_t *out_mask_host_or_null, size_t chunk_cells);
/* BufferOps::min_max (src/buffer.rs:169-173; masked: src/masked/masked_buffer.rs:208-217) of the whole raster:
 * ec_min_max_keys per shard, one all-reduce(MAX) of the 16-byte keys, decode.  masks_or_null == NULL: unmasked.
 * Synchronous result. */
ec_status ec_sharded_min_max(ec_shard_group *g, ec_dtype t, const void *const *p, const uint8_t *const *masks_or_null,
                             const size_t *n, ec_value *mn, ec_value *mx);
/* min_max of an expression program's result over the whole sharded raster, without the raster: ec_expr_min_max_keys per
 * shard, one all-reduce(MAX) of the 16-byte keys, decode (Float64).  masks_or_null[k]: stream k's per-shard masks.
 * Synchronous result. */
ec_status ec_sharded_expr_min_max(ec_shard_group *g, const ec_dtype *dt, const void *const *const *p,
                                  const uint8_t *const *const *masks_or_null, int32_t n_streams, const ec_value *scalars,
                                  int32_t n_scalars, const ec_expr_step *steps, int32_t n_steps, const size_t *n,
                                  ec_value *mn, ec_value *mx);
/* Mask::counts (src/masked/mask.rs:72-80) of the whole raster: per-shard counts, all-reduce(SUM). Synchronous result. */
ec_status ec_sharded_counts(ec_shard_group *g, const uint8_t *const *masks, const size_t *n,
                            uint64_t *n_true, uint64_t *n_false);

/* ---------------------------------------------------------------- *
 * Ordering / equality of whole buffers, decided on the device (no download of the cells).
 * ---------------------------------------------------------------- */
/* impl Ord / PartialEq for CellBuffer — src/buffer.rs:373-436: cell type first, then the first
 * differing cell under the total order (total_cmp for floats; equality is bit equality), then
 * length.  *ordering = -1 / 0 / +1.  Also serves `Mask` (derived Ord on Vec<bool>, mask.rs:10) with
 * lt == rt == EC_U8.  Synchronous result. */
ec_status ec_buffer_cmp(ec_dtype lt, const void *l, size_t nl, ec_dtype rt, const void *r, size_t nr,
                        int32_t *ordering, ec_stream stream);
/* Index of the first cell whose bits differ in l[0..n) vs r[0..n), or n if none. Synchronous result. */
ec_status ec_first_difference(ec_dtype t, const void *l, const void *r, size_t n, uint64_t *index, ec_stream stream);

/* ---------------------------------------------------------------- *
 * Masks.
 * ---------------------------------------------------------------- */
/* MaskedCellBuffer::from_vec_with_nodata — src/masked/masked_buffer.rs:62-71:
 * mask[i] = !(x[i] == nd) under total-order (bitwise) equality, src/masked/nodata.rs:42-49.
 * nd == NULL is NoData::None (all true). nd->dtype must equal t. */
ec_status ec_mask_from_nodata(ec_dtype t, const void *p, size_t n, const ec_value *nd_or_null,
                              uint8_t *mask, ec_stream stream);
/* MaskedCellBuffer::to_vec_with_nodata — src/masked/masked_buffer.rs:143-151:
 * out[i] = mask[i] ? p[i] : nd ; nd == NULL copies p. */
ec_status ec_mask_select(ec_dtype t, const void *p, const uint8_t *mask, size_t n,
                         const ec_value *nd_or_null, void *out, ec_stream stream);
ec_status ec_mask_and(const uint8_t *l, const uint8_t *r, size_t n, uint8_t *out, ec_stream stream); /* mask.rs:118-140 */
ec_status ec_mask_or(const uint8_t *l, const uint8_t *r, size_t n, uint8_t *out, ec_stream stream);  /* mask.rs:142-163 */
ec_status ec_mask_not(const uint8_t *m, size_t n, uint8_t *out, ec_stream stream);                   /* mask.rs:103-116 */
/* Mask::counts — src/masked/mask.rs:72-80. Synchronous result. */
ec_status ec_mask_counts(const uint8_t *m, size_t n, uint64_t *n_true, uint64_t *n_false, ec_stream stream);
/* Asynchronous: {n_true, n_false} to DEVICE memory (SUM all-reduce across shards). */
ec_status ec_mask_counts_device(const uint8_t *m, size_t n, uint64_t *counts2_dev, ec_stream stream);

/* ---------------------------------------------------------------- *
 * Test/bench support (not part of the reference surface).
 * ---------------------------------------------------------------- */
/* x[i] = lo + splitmix64(seed ^ (base + i)) % (hi - lo + 1), t in {EC_U8, EC_U16, EC_F32}
 * (for EC_F32: uniform on [lo, hi) as float from the same stream). */
ec_status ec_synth_fill(ec_dtype t, void *dst, size_t n, uint64_t seed, uint64_t base,
                        double lo, double hi, ec_stream stream);
/* mask[i] = splitmix64(seed ^ (base+i)) % 100 >= pct_nodata */
ec_status ec_synth_mask(uint8_t *dst, size_t n, uint64_t seed, uint64_t base, uint32_t pct_nodata, ec_stream stream);
/* Tuning knobs (each an atomic word: may be set while other host threads launch): "pool_keep_mb" (release threshold
 * of the library's stream-ordered pools), "fused_mixed" (1, default: one-pass kernels for fused chains over mixed
 * cell types; 0: convert to the union type first), "binop_variant" (-1, default: by rule — LDS-staged for an 8-byte operand against one of <= 4 bytes on large rasters; 0 always direct narrow loads; 1 LDS-staged wherever an operand can be staged), "reduce_bpc" (workgroups per CU for reductions; 0, default: as many as are resident at once), "reduce_shape" (A/B launch shapes of min_max, 0 default),
 * "map_u" (16-B groups per lane per tile of the map kernels: 1, 2 or 4), "unaligned_vector" (1, default: vector
 * kernels at any cell offset via unaligned global access; 0: pointers that are not 16-byte aligned run the
 * one-cell-per-lane kernels), "peel" (leading-cell peel of the binop/fused kernels at odd offsets: 0 off, 1 for
 * 1-byte operands (default), 2 also for 2-byte operands).  Measurement knobs (defaults are what ships): "mall_mb",
 * "cache_force", "expr_jit", "expr_fixed", "counts_one_launch", and the occupancy caps "write_lds_kb", "binop_lds_kb", "scalar_lds_kb",
 * "map_lds_kb", "fused_lds_kb" (KiB of unused LDS reserved per workgroup of a kernel family). */
ec_status ec_tune_set(const char *key, int64_t value);
/* Counters for tests: "pool_allocs" (ec_alloc_async calls so far, including those the library makes itself — a call
 * that leaves it unchanged allocated nothing), "devices" (initialised devices), "scratch_streams" (streams the
 * library currently holds reduction scratch for), "binop_lds_rule_launches", "expr_fixed_launches", "expr_interp_launches",
 * "expr_jit_launches" (which kernel form a call took); "tune.<knob>": the current value of a knob of ec_tune_set (so that a
 * scope that turns one can put the previous value back). */
ec_status ec_stat_get(const char *key, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* ERASED_CELLS_H */

// ec_hostpipe.hip — host memory in, host memory out: an expression program streamed through the GPU.
//
// The reference's operands and results are `Vec`s in host memory (src/buffer.rs:12-55); a caller that keeps nothing
// resident pays PCIe both ways — Σ sizeof(T) bytes per cell up and 8 bytes per cell down — and that, not the kernel, is
// the bound (≈ 55 GB/s each way on this box: ≤ 7 Gcells/s for any f64-result operator, against 450-650 Gcells/s resident).
// What the library can do is keep both directions of the link busy at once: ec_host_expr cuts the operands into chunks and
// runs upload, kernel and download on three streams over double-buffered device staging, ordered by events.
//
// Page-locked host memory is what makes the copies asynchronous: ec_host_alloc / ec_host_free hand such memory out, and
// buffers that are not page-locked yet are registered (hipHostRegister) for the duration of the call — pinning pages costs
// about as much as one pass over them, so a caller who reuses its buffers should allocate them with ec_host_alloc.  If
// registration is refused the copies go through the runtime's pageable path (correct, slower).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <vector>

#include "ec_hostpipe.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"
#include "erased_cells.h"

using namespace ecd;

extern "C" ec_status ec_host_alloc(void** hptr, size_t bytes) {
    if (!hptr) return set_error(EC_ERR_ARG, "ec_host_alloc: null out pointer");
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    *hptr = nullptr;
    if (bytes == 0) return EC_OK;
    return check_hip(hipHostMalloc(hptr, bytes, hipHostMallocPortable), "hipHostMalloc");  // page-locked for every device
}

extern "C" ec_status ec_host_free(void* hptr) { return hptr ? check_hip(hipHostFree(hptr), "hipHostFree") : EC_OK; }

namespace {
struct PinTable {
    struct Entry { size_t bytes; int refs; bool registered; bool refused; };  // refused: pin_all asked for a registration and the runtime said no
    std::mutex mu;
    std::condition_variable cv;
    std::multimap<uintptr_t, Entry> live;  // by (page-aligned) base: registrations made here, and ranges copied without one
} g_pins;
constexpr uintptr_t kPage = 4096;

using Interval = std::pair<uintptr_t, uintptr_t>;

// whole pages, merged
std::vector<Interval> merged_pages(const std::vector<std::pair<const void*, size_t>>& ranges) {
    std::vector<Interval> iv;
    for (const auto& r : ranges) {
        if (!r.first || r.second == 0) continue;
        const uintptr_t lo = reinterpret_cast<uintptr_t>(r.first) / kPage * kPage;
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(r.first) + r.second + kPage - 1) / kPage * kPage;
        iv.emplace_back(lo, hi);
    }
    std::sort(iv.begin(), iv.end());
    std::vector<Interval> merged;
    for (const auto& x : iv) {
        if (!merged.empty() && x.first <= merged.back().second) merged.back().second = std::max(merged.back().second, x.second);
        else merged.push_back(x);
    }
    return merged;
}

// does any of `merged` collide with an entry of the table?  A registration collides when it overlaps without covering; a range in
// use without a registration collides with a new REGISTRATION that overlaps it at all (two unregistered users never collide) —
// unless it is a range whose registration the runtime REFUSED and it covers the request: asking again would be refused again, so
// the newcomer shares the entry and copies through the pageable path too.  That is what a shard's pipeline finds when
// ec_sharded_host_expr could not page-lock the whole arrays (a locked-memory limit, a mapping the driver will not pin): before round 4 every shard waited here for
// its own caller, forever.
bool collides(const std::vector<Interval>& merged, bool want_registration) {
    for (const auto& m : merged)
        for (const auto& kv : g_pins.live) {
            const uintptr_t elo = kv.first, ehi = kv.first + kv.second.bytes;
            if (ehi <= m.first || m.second <= elo) continue;
            const bool covers = elo <= m.first && m.second <= ehi;
            if (kv.second.registered) {
                if (!covers) return true;
            } else if (want_registration && !(kv.second.refused && covers)) {
                return true;
            }
        }
    return false;
}

// an entry in flight that covers [lo, hi) and may be shared: a registration, or (for pin_all) a range whose registration was refused
std::multimap<uintptr_t, PinTable::Entry>::iterator covering(const Interval& m, bool or_refused = false) {
    for (auto it = g_pins.live.begin(); it != g_pins.live.end(); ++it)
        if ((it->second.registered || (or_refused && it->second.refused)) && it->first <= m.first && m.second <= it->first + it->second.bytes) return it;
    return g_pins.live.end();
}

// is [p, p + bytes) page-locked memory of the caller's own (ec_host_alloc / hipHostMalloc / its own hipHostRegister)?  Both ends
// are asked: a pageable array may begin right behind a page-locked one.
bool caller_pinned(const void* p, size_t bytes) {
    for (const char* q : {static_cast<const char*>(p), static_cast<const char*>(p) + bytes - 1}) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, q) != hipSuccess || at.type != hipMemoryTypeHost) {
            (void)hipGetLastError();  // "not a HIP pointer" is an answer, not a failure
            return false;
        }
    }
    return true;
}
}  // namespace

namespace ecd {

void PinSet::pin_all(const std::vector<std::pair<const void*, size_t>>& ranges) {
    // the caller's own page-locked arrays need nothing and are left out BEFORE the ranges are merged: judged per array, not per
    // merged interval — an interval that begins in page-locked memory may run on into a pageable neighbour, whose copies are
    // exactly what the table is there to guard
    // — and judged UNDER the table's lock, against the table first: pages another call of the library has registered for the time being
    // report as page-locked host memory too, and a range taken for the caller's own on that evidence would hold no reference when that
    // call unregisters them (hipMemcpyAsync then fails with "invalid argument": the six-thread soak found this ordering within seconds)
    const bool refuse_all = tuning().inject_pin_refusal.load() != 0;  // test hook: every registration "refused"
    std::unique_lock<std::mutex> lk(g_pins.mu);
    std::vector<Interval> merged;
    for (;;) {
        std::vector<std::pair<const void*, size_t>> mine;
        for (const auto& r : ranges) {
            if (!r.first || !r.second) continue;
            const uintptr_t lo = reinterpret_cast<uintptr_t>(r.first), hi = lo + r.second;
            bool ours = false;  // touches a registration the library made
            for (const auto& kv : g_pins.live)
                ours = ours || (kv.second.registered && kv.first < hi && lo < kv.first + kv.second.bytes);
            if (ours || !caller_pinned(r.first, r.second)) mine.push_back(r);
        }
        merged = merged_pages(mine);
        if (!collides(merged, true)) break;
        g_pins.cv.wait(lk);  // holding nothing while it waits: all of a call's ranges are taken at once
    }
    for (const auto& m : merged) {
        auto it = covering(m, true);
        if (it != g_pins.live.end()) {  // inside a registration in flight, or inside a range that could not be registered: share it
            ++it->second.refs;
            held_.push_back(Held{it->first, it->second.bytes, it->second.registered, it->second.refused});
            continue;
        }
        // page-locked for every device: a shard group copies from it on all of them
        if (!refuse_all && hipHostRegister(reinterpret_cast<void*>(m.first), m.second - m.first, hipHostRegisterPortable) == hipSuccess) {
            g_pins.live.emplace(m.first, PinTable::Entry{m.second - m.first, 1, true, false});
            held_.push_back(Held{m.first, m.second - m.first, true});
        } else {
            (void)hipGetLastError();  // refused (RLIMIT_MEMLOCK, a mapping the driver will not pin, ...): the runtime's pageable path copies such a range —
            g_pins.live.emplace(m.first, PinTable::Entry{m.second - m.first, 1, false, true});  // — which makes it a range in use
            held_.push_back(Held{m.first, m.second - m.first, false, true});
        }
    }
}

void PinSet::use_all(const std::vector<std::pair<const void*, size_t>>& ranges) {
    const std::vector<Interval> merged = merged_pages(ranges);
    std::unique_lock<std::mutex> lk(g_pins.mu);
    while (collides(merged, false)) g_pins.cv.wait(lk);
    for (const auto& m : merged) {
        auto it = covering(m);
        if (it != g_pins.live.end()) {
            ++it->second.refs;
            held_.push_back(Held{it->first, it->second.bytes, true});
        } else {
            g_pins.live.emplace(m.first, PinTable::Entry{m.second - m.first, 1, false, false});
            held_.push_back(Held{m.first, m.second - m.first, false});
        }
    }
}

PinSet::~PinSet() {
    if (held_.empty()) return;
    std::lock_guard<std::mutex> lk(g_pins.mu);
    for (const Held& h : held_) {
        auto range = g_pins.live.equal_range(h.base);
        for (auto it = range.first; it != range.second; ++it) {
            if (it->second.registered != h.registered || it->second.refused != h.refused || it->second.bytes != h.bytes) continue;
            if (--it->second.refs == 0) {
                if (h.registered) (void)hipHostUnregister(reinterpret_cast<void*>(h.base));
                g_pins.live.erase(it);
            }
            break;
        }
    }
    g_pins.cv.notify_all();
}

}  // namespace ecd

namespace {

struct Pipe {
    hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_cmp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    void* stage[2] = {nullptr, nullptr};
    bool simple = false;  // one chunk on the calling thread's own stream (hipStreamPerThread): no streams or events to create
    ec_status open_simple(size_t stage_bytes) {
        simple = true;
        s_in = s_cmp = s_out = hipStreamPerThread;
        return ec_alloc_async(&stage[0], stage_bytes, s_cmp);  // used in stream order on the same stream
    }
    ec_status open(size_t stage_bytes) {
        for (hipStream_t* s : {&s_in, &s_cmp, &s_out}) {
            ec_status st = check_hip(hipStreamCreateWithFlags(s, hipStreamNonBlocking), "hipStreamCreateWithFlags");
            if (st != EC_OK) return st;
        }
        for (int k = 0; k < 2; ++k) {
            for (hipEvent_t* e : {&ev_in[k], &ev_cmp[k], &ev_out[k]}) {
                ec_status st = check_hip(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreateWithFlags");
                if (st != EC_OK) return st;
            }
            // from the library's stream-ordered pool: a second call finds the blocks of the first (hipMalloc + hipFree of two
            // 370 MB slots cost ≈ 10 ms a call, a sixth of a 16384² divide's transfer time)
            ec_status st = ec_alloc_async(&stage[k], stage_bytes, s_cmp);
            if (st != EC_OK) return st;
        }
        return check_hip(hipStreamSynchronize(s_cmp), "hipStreamSynchronize");  // the blocks exist before the copy streams touch them
    }
    ~Pipe() {
        if (simple) {
            (void)hipStreamSynchronize(s_cmp);
            if (stage[0]) (void)ec_free_async(stage[0], s_cmp);
            return;
        }
        for (hipStream_t s : {s_in, s_out, s_cmp})
            if (s) (void)hipStreamSynchronize(s);
        for (int k = 0; k < 2; ++k)
            if (stage[k] && s_cmp) (void)ec_free_async(stage[k], s_cmp);
        for (hipStream_t s : {s_in, s_cmp, s_out})
            if (s) {
                (void)hipStreamSynchronize(s);
                (void)hipStreamDestroy(s);
            }
        for (int k = 0; k < 2; ++k)
            for (hipEvent_t e : {ev_in[k], ev_cmp[k], ev_out[k]})
                if (e) (void)hipEventDestroy(e);
    }
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
constexpr size_t kRegisterFromBytes = size_t(64) << 20;  // link traffic from which a call page-locks the caller's arrays (below: pageable copies)

}  // namespace

// The pipeline behind ec_host_expr and ec_host_masked_expr.  Masked form (nodata != nullptr): per chunk, on the device,
// mask_k = (stream k != nodata[k]) (from_vec_with_nodata, src/masked/masked_buffer.rs:62-71), the program with the AND of the
// masks (impl $trt for &MaskedCellBuffer, :326-364), then out = mask ? value : *out_nodata (to_vec_with_nodata, :137-152) —
// what travels back is still 8 bytes per cell (+ 1 when the caller wants the mask).
static ec_status host_pipeline(const char* what, const ec_dtype* dt, const void* const* p_host, const ec_value* const* nodata, int32_t n_streams,
                               const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, size_t n, double* out_host,
                               const double* out_nodata, uint8_t* out_mask_host, size_t chunk_cells) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    {   // the program, refused before anything is allocated
        size_t len = 0;
        if ((st = ec_expr_source(dt, n_streams, n_scalars, steps, n_steps, nullptr, nullptr, 0, &len)) != EC_OK) return st;
    }
    const bool masked = nodata != nullptr;
    if (masked)
        for (int k = 0; k < n_streams; ++k)
            if (nodata[k] && nodata[k]->dtype != dt[k])
                return set_error(EC_ERR_ARG, "%s: nodata[%d] has cell type %d, its stream %d", what, k, int(nodata[k]->dtype), int(dt[k]));
    if (n == 0) return EC_OK;
    if (!p_host || !out_host) return set_error(EC_ERR_ARG, "%s: null pointer", what);
    for (int k = 0; k < n_streams; ++k)
        if (!p_host[k]) return set_error(EC_ERR_ARG, "%s: stream %d is null", what, k);
    // A small call is not worth a pipeline: three streams, six events, two staging slots and the page-locking of the caller's
    // buffers cost ≈ 1.1 ms (profiles/r03/host_pipeline.md) — more than they save below ≈ 64 MiB over the link.  Such a call runs as
    // one chunk on the calling thread's own stream, its (pageable) buffers copied by the runtime.
    size_t link_bytes = n * sizeof(double) + (out_mask_host ? n : 0);
    for (int k = 0; k < n_streams; ++k) link_bytes += n * ecl::size_of(dt[k]);
    const bool small = chunk_cells == 0 && link_bytes <= (size_t(64) << 20);
    const size_t chunk = small ? n : std::min(n, chunk_cells ? chunk_cells : size_t(1) << 25);
    // one staging slot: the operands' chunks, the f64 result (and, masked: the streams' masks, the result's mask, the selected
    // result), each on a 256-byte boundary
    size_t off[4], off_out, off_mask[4] = {0, 0, 0, 0}, off_omask = 0, off_sel = 0, bytes_per_cell[4] = {0, 0, 0, 0}, at = 0;
    for (int k = 0; k < n_streams; ++k) {
        bytes_per_cell[k] = ecl::size_of(dt[k]);
        off[k] = at;
        at = align_up(at + chunk * bytes_per_cell[k], 256);
    }
    off_out = at;
    at = align_up(at + chunk * sizeof(double), 256);
    if (masked) {
        for (int k = 0; k < n_streams; ++k) {
            off_mask[k] = at;
            at = align_up(at + chunk, 256);
        }
        off_omask = at;
        at = align_up(at + chunk, 256);
        off_sel = at;
        at = align_up(at + chunk * sizeof(double), 256);
    }
    PinSet pins;
    {
        std::vector<std::pair<const void*, size_t>> ranges;
        for (int k = 0; k < n_streams; ++k) ranges.emplace_back(p_host[k], n * bytes_per_cell[k]);
        ranges.emplace_back(out_host, n * sizeof(double));
        if (out_mask_host) ranges.emplace_back(out_mask_host, n);
        // Page-locking the caller's arrays pays from ≈ 64 MiB over the link — and only such arrays are safe to register: malloc maps them on
        // their own, so their pages hold nothing else.  SMALL arrays share heap pages with other objects, and the HIP runtime pins heap pages
        // itself, lazily and with a cache, whenever somebody copies from or to pageable memory (from_vec / to_vec of a neighbour): a
        // hipHostRegister / hipHostUnregister of the same pages pulls the GPU mapping from under that cache, and a later copy faults on a
        // HOST address ("Memory access fault by GPU … Reason: Unknown", twice in round 4's suite runs, in a pipeline forced onto 31,434-cell
        // arrays with chunk_cells = 2000).  Below the threshold the chunks are copied by the runtime's pageable path, whatever chunk_cells says.
        if (small || link_bytes <= kRegisterFromBytes) pins.use_all(ranges);  // no registration may come or go under the copies
        else pins.pin_all(ranges);
    }
    Pipe pipe;
    if ((st = small ? pipe.open_simple(at) : pipe.open(at)) != EC_OK) return st;
    ec_value sel{};
    if (out_nodata) {
        sel.dtype = EC_F64;
        sel.v.f64 = *out_nodata;
    }
    const size_t nchunks = (n + chunk - 1) / chunk;
    for (size_t c = 0; c < nchunks && st == EC_OK; ++c) {
        const int k = static_cast<int>(c & 1);
        const size_t lo = c * chunk, m = std::min(chunk, n - lo);
        char* slot = static_cast<char*>(pipe.stage[k]);
        // upload: once the kernel that last read this slot's operands has finished
        if (!pipe.simple) st = check_hip(hipStreamWaitEvent(pipe.s_in, pipe.ev_cmp[k], 0), "hipStreamWaitEvent");
        const void* dptr[4] = {nullptr, nullptr, nullptr, nullptr};
        const uint8_t* dmask[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < n_streams && st == EC_OK; ++j) {
            dptr[j] = slot + off[j];
            st = check_hip(hipMemcpyAsync(slot + off[j], static_cast<const char*>(p_host[j]) + lo * bytes_per_cell[j], m * bytes_per_cell[j],
                                          hipMemcpyHostToDevice, pipe.s_in), "hipMemcpyAsync(H2D)");
        }
        if (st == EC_OK && !pipe.simple) st = check_hip(hipEventRecord(pipe.ev_in[k], pipe.s_in), "hipEventRecord");
        // compute: once the operands are there and the slot's previous result has left for the host
        if (st == EC_OK && !pipe.simple) st = check_hip(hipStreamWaitEvent(pipe.s_cmp, pipe.ev_in[k], 0), "hipStreamWaitEvent");
        if (st == EC_OK && !pipe.simple) st = check_hip(hipStreamWaitEvent(pipe.s_cmp, pipe.ev_out[k], 0), "hipStreamWaitEvent");
        double* dout = reinterpret_cast<double*>(slot + off_out);
        const double* result = dout;
        uint8_t* domask = reinterpret_cast<uint8_t*>(slot + off_omask);
        if (!masked) {
            if (st == EC_OK) st = ec_expr(dt, dptr, n_streams, scalars, n_scalars, steps, n_steps, m, dout, pipe.s_cmp);
        } else {
            for (int j = 0; j < n_streams && st == EC_OK; ++j) {
                uint8_t* mk = reinterpret_cast<uint8_t*>(slot + off_mask[j]);
                dmask[j] = mk;
                st = ec_mask_from_nodata(dt[j], dptr[j], m, nodata[j], mk, pipe.s_cmp);
            }
            if (st == EC_OK) st = ec_masked_expr(dt, dptr, dmask, n_streams, scalars, n_scalars, steps, n_steps, m, dout, domask, pipe.s_cmp);
            if (st == EC_OK && out_nodata) {
                double* dsel = reinterpret_cast<double*>(slot + off_sel);
                st = ec_mask_select(EC_F64, dout, domask, m, &sel, dsel, pipe.s_cmp);
                result = dsel;
            }
        }
        if (st == EC_OK && !pipe.simple) st = check_hip(hipEventRecord(pipe.ev_cmp[k], pipe.s_cmp), "hipEventRecord");
        // download
        if (st == EC_OK && !pipe.simple) st = check_hip(hipStreamWaitEvent(pipe.s_out, pipe.ev_cmp[k], 0), "hipStreamWaitEvent");
        if (st == EC_OK) st = check_hip(hipMemcpyAsync(out_host + lo, result, m * sizeof(double), hipMemcpyDeviceToHost, pipe.s_out), "hipMemcpyAsync(D2H)");
        if (st == EC_OK && masked && out_mask_host)
            st = check_hip(hipMemcpyAsync(out_mask_host + lo, domask, m, hipMemcpyDeviceToHost, pipe.s_out), "hipMemcpyAsync(D2H mask)");
        if (st == EC_OK && !pipe.simple) st = check_hip(hipEventRecord(pipe.ev_out[k], pipe.s_out), "hipEventRecord");
    }
    const std::string keep = st != EC_OK ? last_error_text() : std::string();
    for (hipStream_t s : {pipe.s_in, pipe.s_cmp, pipe.s_out}) {
        const ec_status w = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        if (st == EC_OK && w != EC_OK) st = w;
        if (pipe.simple) break;  // one stream
    }
    if (!keep.empty()) return set_error_text(st, keep);
    return st;
}

extern "C" ec_status ec_host_expr(const ec_dtype* dt, const void* const* p_host, int32_t n_streams, const ec_value* scalars, int32_t n_scalars,
                                  const ec_expr_step* steps, int32_t n_steps, size_t n, double* out_host, size_t chunk_cells) {
    return host_pipeline("ec_host_expr", dt, p_host, nullptr, n_streams, scalars, n_scalars, steps, n_steps, n, out_host, nullptr, nullptr, chunk_cells);
}

extern "C" ec_status ec_host_masked_expr(const ec_dtype* dt, const void* const* p_host, const ec_value* const* nodata, int32_t n_streams,
                                         const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, size_t n,
                                         double* out_host, const double* out_nodata_or_null, uint8_t* out_mask_host_or_null, size_t chunk_cells) {
    if (!nodata) return set_error(EC_ERR_ARG, "ec_host_masked_expr: null nodata array (entries may be NULL: NoData::None)");
    return host_pipeline("ec_host_masked_expr", dt, p_host, nodata, n_streams, scalars, n_scalars, steps, n_steps, n, out_host, out_nodata_or_null,
                         out_mask_host_or_null, chunk_cells);
}

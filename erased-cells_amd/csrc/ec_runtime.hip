// ec_runtime.hip — devices, memory, streams, errors and tuning of liberased_cells_hip.so.
//
// The runtime is per DEVICE: every device passed to ec_init() gets its own CU count, its own
// library-owned stream-ordered memory pool and its own per-stream reduction scratch.  A host thread
// works on one device at a time (its "current library device": the one of its last ec_init /
// ec_set_device, by default the first device the process initialised); one process can therefore drive
// one GPU (the one-rank-per-GPU shape of bench.py) or all eight (ec_shard_group_*, ec_sharded.hip).
//
// There is no CPU fallback: without a HIP device every compute entry point returns
// EC_ERR_HIP / EC_ERR_NOT_INITIALIZED.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "ec_hostpipe.hpp"
#include "ec_runtime.hpp"

namespace ecd {

// ------------------------------------------------------------------ errors (thread-local)
static thread_local std::string t_err;
static thread_local int t_narrow_src = -1, t_narrow_dst = -1;

ec_status set_error(ec_status code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_err = buf;
    return code;
}

ec_status set_error_text(ec_status code, const std::string& text) {
    t_err = text;
    return code;
}

const std::string& last_error_text() { return t_err; }

ec_status set_narrowing(int src, int dst) {
    static const char* names[EC_NTYPES] = {"UInt8", "UInt16", "UInt32", "UInt64", "Int8",
                                           "Int16", "Int32", "Int64", "Float32", "Float64"};
    t_narrow_src = src;
    t_narrow_dst = dst;
    // message text of Error::NarrowingError (src/error.rs:14)
    return set_error(EC_ERR_NARROWING, "Invalid narrowing from cell-type %s to %s", names[src], names[dst]);
}

ec_status check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return EC_OK;
    return set_error(EC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

ec_status check_launch(const char* what) { return check_hip(hipGetLastError(), what); }

// ------------------------------------------------------------------ per-device state
struct ScratchEntry {
    Scratch sc;
    uint64_t stamp = 0;  // last use, for eviction
};

}  // namespace ecd (reopened below: ScratchOwner is named in ec_runtime.hpp)

// Owns one stream's scratch allocations.  Destroyed when the table's entry AND every copy a caller still holds are gone —
// on whatever thread drops the last one, never while the runtime's table lock is held (hipFree waits for the device).
struct ecd::ScratchOwner {
    int device = -1;
    int64_t* dev = nullptr;
    int64_t* host = nullptr;
    std::mutex* mu = nullptr;
    ~ScratchOwner() {
        int before = -1;
        const bool had = hipGetDevice(&before) == hipSuccess;
        if (device >= 0 && (!had || before != device)) (void)hipSetDevice(device);
        (void)hipFree(dev);
        (void)hipHostFree(host);
        delete mu;
        if (had && before != device) (void)hipSetDevice(before);
        (void)hipGetLastError();
    }
};

namespace ecd {

struct DeviceState {
    int device = -1;
    int cus = 256;
    hipMemPool_t pool = nullptr;  // library-owned; nullptr = creation failed, the device's default pool serves (untouched)
    std::map<hipStream_t, ScratchEntry> scratch;
};

// A foreign stream (torch's, a caller's) gets scratch at first use and nobody tells the library when it dies, so the
// table is bounded: beyond kMaxScratchPerDevice streams the least recently used entry is released (hipFree waits for
// the device, so no kernel can still be writing it; a stream used again later simply gets a fresh entry).
constexpr size_t kMaxScratchPerDevice = 64;

static std::mutex g_mu;
static std::map<int, DeviceState> g_devs;
static std::atomic<int> g_default_device{-1};
static std::atomic<uint64_t> g_generation{1};  // bumped by ec_shutdown: invalidates every thread's cached binding
static std::atomic<uint64_t> g_stamp{0};
static std::atomic<int64_t> g_lds_rule_launches{0};
std::atomic<int64_t>& lds_rule_launches() { return g_lds_rule_launches; }
static std::atomic<int64_t> g_pool_allocs{0};  // ec_alloc_async calls that reached the pool (ec_stat_get)
static Tuning g_tuning;

static thread_local int t_device = -1;          // this thread's choice (ec_init / ec_set_device), -1 = process default
static thread_local uint64_t t_generation = 0;
static thread_local int t_active = -1;          // device bound by the last ensure_ready()
static thread_local int t_cus = 256;

int64_t expr_stat(const char* key, bool* known);  // ec_expr.hip
void expr_jit_release();                             // ec_expr_jit.hip
Tuning& tuning() { return g_tuning; }
int device_cus() { return t_cus; }
int current_device() { return t_active; }

// (freeing a stream's scratch = dropping the table's reference: ScratchOwner's destructor does the work once nobody uses it)

ec_status ensure_ready() {
    if (t_generation != g_generation.load(std::memory_order_acquire)) {
        t_device = -1;
        t_active = -1;
        t_generation = g_generation.load(std::memory_order_acquire);
    }
    const int dev = t_device >= 0 ? t_device : g_default_device.load(std::memory_order_acquire);
    if (dev < 0) return set_error(EC_ERR_NOT_INITIALIZED, "ec_init() has not been called (no HIP device bound)");
    // HIP's current device is per host thread, and other code in the process (torch, the caller) may move it
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != dev) {
        ec_status st = check_hip(hipSetDevice(dev), "hipSetDevice");
        if (st != EC_OK) return st;
    }
    if (t_active != dev) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_devs.find(dev);
        if (it == g_devs.end()) return set_error(EC_ERR_NOT_INITIALIZED, "device %d is not initialised (ec_init)", dev);
        t_cus = it->second.cus;
        t_active = dev;
    }
    return EC_OK;
}

ec_status get_scratch(hipStream_t s, Scratch* out) {
    std::shared_ptr<ScratchOwner> evicted;  // destroyed after the lock is released
    std::lock_guard<std::mutex> lk(g_mu);
    auto dit = g_devs.find(t_active);
    if (dit == g_devs.end()) return set_error(EC_ERR_NOT_INITIALIZED, "get_scratch: no device bound");
    auto& table = dit->second.scratch;
    auto it = table.find(s);
    if (it == table.end()) {
        if (table.size() >= kMaxScratchPerDevice) {
            // least recently used entry that nobody is inside of (only the table references it)
            auto victim = table.end();
            for (auto k = table.begin(); k != table.end(); ++k)
                if (k->second.sc.owner.use_count() == 1 && (victim == table.end() || k->second.stamp < victim->second.stamp)) victim = k;
            if (victim != table.end()) {
                evicted = std::move(victim->second.sc.owner);
                table.erase(victim);
            }  // else: every entry is in use right now — the table grows past its bound for the moment
        }
        ScratchEntry e;
        auto own = std::make_shared<ScratchOwner>();
        own->device = t_active;
        ec_status st = check_hip(hipMalloc(reinterpret_cast<void**>(&own->dev), (2 * kMaxReduceBlocks + 8) * sizeof(int64_t)),
                                 "hipMalloc(scratch)");
        if (st != EC_OK) return st;
        // the accumulator words (Scratch::dev_acc) start at zero; every kernel that uses one leaves it at zero again
        st = check_hip(hipMemset(own->dev + 2 * kMaxReduceBlocks, 0, 8 * sizeof(int64_t)), "hipMemset(scratch)");
        if (st != EC_OK) return st;
        st = check_hip(hipHostMalloc(reinterpret_cast<void**>(&own->host), 4 * sizeof(int64_t), hipHostMallocDefault),
                       "hipHostMalloc(scratch)");
        if (st != EC_OK) return st;  // `own` frees what it has
        own->mu = new std::mutex;
        e.sc.dev = own->dev;
        e.sc.host = own->host;
        e.sc.mu = own->mu;
        void* as_device = nullptr;
        static const bool no_zero_copy = std::getenv("EC_NO_ZERO_COPY_RESULTS") != nullptr;  // A/B switch
        if (!no_zero_copy && hipHostGetDevicePointer(&as_device, e.sc.host, 0) == hipSuccess && as_device) e.sc.host_dev = static_cast<int64_t*>(as_device);
        else (void)hipGetLastError();  // host_dev stays null: results go through dev_result() and a copy
        e.sc.owner = std::move(own);
        it = table.emplace(s, e).first;
    }
    it->second.stamp = ++g_stamp;
    *out = it->second.sc;  // shares ownership
    return EC_OK;
}

static void set_pool_threshold(hipMemPool_t pool) {
    uint64_t keep = static_cast<uint64_t>(g_tuning.pool_keep_mb.load()) << 20;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
}

static inline hipStream_t S(ec_stream s) { return static_cast<hipStream_t>(s); }

}  // namespace ecd

using namespace ecd;

// =================================================================== runtime
extern "C" int32_t ec_abi_version(void) { return EC_ABI_VERSION; }

extern "C" ec_status ec_init(int32_t device) {
    std::lock_guard<std::mutex> lk(g_mu);
    t_generation = g_generation.load();
    if (g_devs.count(device)) {  // idempotent; also (re)binds the calling thread
        t_device = device;
        t_active = -1;
        return EC_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_error(EC_ERR_HIP, "ec_init: no HIP device (%s)", e == hipSuccess ? "count == 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return set_error(EC_ERR_ARG, "ec_init: device %d of %d", device, count);
    ec_status st = check_hip(hipSetDevice(device), "hipSetDevice");
    if (st != EC_OK) return st;
    hipDeviceProp_t prop;
    st = check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (st != EC_OK) return st;
    DeviceState ds;
    ds.device = device;
    ds.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // Stream-ordered allocations (ec_alloc_async) come from a pool this library owns — the device's default pool,
    // which other code in the process may use, is left alone.  Freed blocks stay cached up to the release
    // threshold ("pool_keep_mb") instead of going back to the driver at every synchronisation.
    hipMemPoolProps pp;
    std::memset(&pp, 0, sizeof pp);
    pp.allocType = hipMemAllocationTypePinned;
    pp.handleTypes = hipMemHandleTypeNone;
    pp.location.type = hipMemLocationTypeDevice;
    pp.location.id = device;
    if (hipMemPoolCreate(&ds.pool, &pp) == hipSuccess) set_pool_threshold(ds.pool);
    else ds.pool = nullptr;
    (void)hipGetLastError();
    g_devs.emplace(device, ds);
    int none = -1;
    g_default_device.compare_exchange_strong(none, device);
    t_device = device;
    t_active = -1;
    return EC_OK;  // (reduction scratch of a stream is created by ec_prepare_stream / first use)
}

extern "C" ec_status ec_set_device(int32_t device) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_devs.count(device)) return set_error(EC_ERR_NOT_INITIALIZED, "ec_set_device: device %d is not initialised (ec_init)", device);
    }
    t_generation = g_generation.load();
    t_device = device;
    return ensure_ready();
}

extern "C" ec_status ec_get_device(int32_t* device) {
    if (!device) return set_error(EC_ERR_ARG, "ec_get_device: null out");
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    *device = t_active;
    return EC_OK;
}

extern "C" ec_status ec_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    int before = -1;
    const bool had = !g_devs.empty() && hipGetDevice(&before) == hipSuccess;
    if (!g_devs.empty()) expr_jit_release();  // modules of run-time compiled expression programs (ec_expr_jit.hip)
    for (auto& kv : g_devs) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        kv.second.scratch.clear();  // no call is in flight at shutdown: the owners free their memory here
        if (kv.second.pool) {
            (void)hipMemPoolTrimTo(kv.second.pool, 0);
            (void)hipMemPoolDestroy(kv.second.pool);
        }
    }
    if (had) (void)hipSetDevice(before);  // the caller's HIP device is left as it was
    (void)hipGetLastError();
    g_devs.clear();
    g_default_device.store(-1);
    g_generation.fetch_add(1);
    t_device = t_active = -1;
    return EC_OK;
}

extern "C" const char* ec_last_error_string(void) { return t_err.c_str(); }

extern "C" ec_status ec_last_narrowing(ec_dtype* src, ec_dtype* dst) {
    if (!src || !dst || t_narrow_src < 0) return set_error(EC_ERR_ARG, "ec_last_narrowing: nothing recorded");
    *src = static_cast<ec_dtype>(t_narrow_src);
    *dst = static_cast<ec_dtype>(t_narrow_dst);
    return EC_OK;
}

extern "C" ec_status ec_device_info(int32_t* n_cu, uint64_t* hbm_bytes, char* name, size_t name_cap) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    hipDeviceProp_t prop;
    st = check_hip(hipGetDeviceProperties(&prop, t_active), "hipGetDeviceProperties");
    if (st != EC_OK) return st;
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (name && name_cap) { std::strncpy(name, prop.gcnArchName, name_cap - 1); name[name_cap - 1] = 0; }
    return EC_OK;
}

// ------------------------------------------------------------------ memory
extern "C" ec_status ec_alloc(void** dptr, size_t bytes) {
    if (!dptr) return set_error(EC_ERR_ARG, "ec_alloc: null out pointer");
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    *dptr = nullptr;
    if (bytes == 0) return EC_OK;
    return check_hip(hipMalloc(dptr, bytes), "hipMalloc");
}
extern "C" ec_status ec_free(void* dptr) { return dptr ? check_hip(hipFree(dptr), "hipFree") : EC_OK; }

// Result buffers of eager operators are allocated per call (the reference `collect()`s a fresh Vec);
// hipMalloc/hipFree cost 0.2-0.4 ms per pair — as much as the 16384² kernel itself — and hipFree
// synchronises.  The stream-ordered pool makes both a queue operation (tools/alloc_cost.py).
extern "C" ec_status ec_alloc_async(void** dptr, size_t bytes, ec_stream stream) {
    if (!dptr) return set_error(EC_ERR_ARG, "ec_alloc_async: null out pointer");
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    *dptr = nullptr;
    if (bytes == 0) return EC_OK;
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_devs.find(t_active);
        if (it != g_devs.end()) pool = it->second.pool;
    }
    g_pool_allocs.fetch_add(1, std::memory_order_relaxed);
    if (pool) return check_hip(hipMallocFromPoolAsync(dptr, bytes, pool, S(stream)), "hipMallocFromPoolAsync");
    return check_hip(hipMallocAsync(dptr, bytes, S(stream)), "hipMallocAsync");
}
extern "C" ec_status ec_free_async(void* dptr, ec_stream stream) {
    return dptr ? check_hip(hipFreeAsync(dptr, S(stream)), "hipFreeAsync") : EC_OK;
}

// Free of a block whose last use may be on another stream than the one it is returned on: the free is ordered
// after everything enqueued so far on `last_use_stream` (event + wait), then queued on `alloc_stream`.  The calling
// thread may be bound to another device than the block's (a destructor run by a garbage collector, a thread that
// moved on with ec_set_device): the block's own device is looked up and made current for the duration.  If the
// event path fails the free still happens, behind a wait for `last_use_stream` — a block is never leaked.
extern "C" ec_status ec_free_ordered(void* dptr, ec_stream alloc_stream, ec_stream last_use_stream) {
    if (!dptr) return EC_OK;
    int before = -1, owner = -1;
    const bool had = hipGetDevice(&before) == hipSuccess;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dptr) == hipSuccess) owner = attr.device;
    else (void)hipGetLastError();
    const bool moved = owner >= 0 && had && owner != before && hipSetDevice(owner) == hipSuccess;
    ec_status st = EC_OK;
    if (alloc_stream != last_use_stream) {
        hipEvent_t ev = nullptr;
        st = check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        if (st == EC_OK) st = check_hip(hipEventRecord(ev, S(last_use_stream)), "hipEventRecord");
        if (st == EC_OK) st = check_hip(hipStreamWaitEvent(S(alloc_stream), ev, 0), "hipStreamWaitEvent");
        if (ev) (void)hipEventDestroy(ev);  // released once the wait has consumed it
        if (st != EC_OK) {  // order the hard way instead of leaking the block
            (void)hipGetLastError();
            (void)hipStreamSynchronize(S(last_use_stream));
        }
    }
    const ec_status freed = check_hip(hipFreeAsync(dptr, S(alloc_stream)), "hipFreeAsync");
    if (moved) (void)hipSetDevice(before);
    return freed != EC_OK ? freed : EC_OK;
}

extern "C" ec_status ec_pool_trim(size_t keep_bytes) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_devs.find(t_active);
        if (it != g_devs.end()) pool = it->second.pool;
    }
    if (!pool) return EC_OK;
    st = check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (st != EC_OK) return st;
    return check_hip(hipMemPoolTrimTo(pool, keep_bytes), "hipMemPoolTrimTo");
}

extern "C" ec_status ec_upload(void* dst_dev, const void* src_host, size_t bytes, ec_stream stream) {
    if (bytes == 0) return EC_OK;
    if (!dst_dev || !src_host) return set_error(EC_ERR_ARG, "ec_upload: null pointer");
    PinSet in_use;  // no page-lock registration of this range (a host-to-host call in another thread) may come or go under the copy
    in_use.use_all({{src_host, bytes}});
    ec_status st = check_hip(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, S(stream)), "hipMemcpyAsync(H2D)");
    if (st != EC_OK) return st;
    return check_hip(hipStreamSynchronize(S(stream)), "hipStreamSynchronize");  // src_host may be pageable
}
extern "C" ec_status ec_download(void* dst_host, const void* src_dev, size_t bytes, ec_stream stream) {
    if (bytes == 0) return EC_OK;
    if (!dst_host || !src_dev) return set_error(EC_ERR_ARG, "ec_download: null pointer");
    PinSet in_use;  // (as in ec_upload)
    in_use.use_all({{dst_host, bytes}});
    ec_status st = check_hip(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, S(stream)), "hipMemcpyAsync(D2H)");
    if (st != EC_OK) return st;
    return check_hip(hipStreamSynchronize(S(stream)), "hipStreamSynchronize");
}
extern "C" ec_status ec_copy(void* dst_dev, const void* src_dev, size_t bytes, ec_stream stream) {
    if (bytes == 0) return EC_OK;
    if (!dst_dev || !src_dev) return set_error(EC_ERR_ARG, "ec_copy: null pointer");
    return check_hip(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, S(stream)), "hipMemcpyAsync(D2D)");
}

// ------------------------------------------------------------------ streams
extern "C" ec_status ec_stream_create(ec_stream* out) {
    if (!out) return set_error(EC_ERR_ARG, "ec_stream_create: null out");
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    hipStream_t s;
    st = check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    if (st != EC_OK) return st;
    *out = s;
    return ec_prepare_stream(s);
}

extern "C" ec_status ec_prepare_stream(ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    Scratch sc;
    return get_scratch(S(stream), &sc);  // allocates this stream's reduction scratch now, not at first use
}

// Releases the reduction scratch the library holds for `stream` (any stream, also one it did not create).  The memory
// goes once no call is using it any more.  A stream with reductions captured in a hipGraph must not be released while
// the graph may still be replayed: the graph's kernels carry the scratch addresses.
extern "C" ec_status ec_release_stream(ec_stream stream) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    std::shared_ptr<ScratchOwner> released;  // destroyed after the lock is released
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto dit = g_devs.find(t_active);
        if (dit == g_devs.end()) return EC_OK;
        auto it = dit->second.scratch.find(S(stream));
        if (it != dit->second.scratch.end()) {
            released = std::move(it->second.sc.owner);
            dit->second.scratch.erase(it);
        }
    }
    return EC_OK;
}

extern "C" ec_status ec_stream_destroy(ec_stream s) {
    ec_status st = ec_release_stream(s);
    if (st != EC_OK) return st;
    return check_hip(hipStreamDestroy(S(s)), "hipStreamDestroy");
}
extern "C" ec_status ec_stream_sync(ec_stream s) { return check_hip(hipStreamSynchronize(S(s)), "hipStreamSynchronize"); }

// ------------------------------------------------------------------ statistics
extern "C" ec_status ec_stat_get(const char* key, int64_t* value) {
    if (!key || !value) return set_error(EC_ERR_ARG, "ec_stat_get: null argument");
    if (!std::strcmp(key, "pool_allocs")) *value = g_pool_allocs.load(std::memory_order_relaxed);
    else if (!std::strcmp(key, "binop_lds_rule_launches")) *value = g_lds_rule_launches.load(std::memory_order_relaxed);
    else if (!std::strncmp(key, "tune.", 5)) {  // the current value of a knob of ec_tune_set, so that a caller can put it back
        const char* k = key + 5;
        if (!std::strcmp(k, "binop_variant")) *value = g_tuning.binop_variant;
        else if (!std::strcmp(k, "reduce_bpc")) *value = g_tuning.reduce_bpc;
        else if (!std::strcmp(k, "reduce_shape")) *value = g_tuning.reduce_shape;
        else if (!std::strcmp(k, "map_u")) *value = g_tuning.map_u;
        else if (!std::strcmp(k, "peel")) *value = g_tuning.peel;
        else if (!std::strcmp(k, "unaligned_vector")) *value = g_tuning.unaligned_vector;
        else if (!std::strcmp(k, "fused_mixed")) *value = g_tuning.fused_mixed;
        else if (!std::strcmp(k, "mall_mb")) *value = g_tuning.mall_mb;
        else if (!std::strcmp(k, "expr_jit")) *value = g_tuning.expr_jit;
        else if (!std::strcmp(k, "expr_fixed")) *value = g_tuning.expr_fixed;
        else if (!std::strcmp(k, "write_lds_kb")) *value = g_tuning.write_lds_kb;
        else if (!std::strcmp(k, "fused_lds_kb")) *value = g_tuning.fused_lds_kb;
        else if (!std::strcmp(k, "counts_one_launch")) *value = g_tuning.counts_one_launch;
        else if (!std::strcmp(k, "cache_force")) *value = g_tuning.cache_force;
        else if (!std::strcmp(k, "pool_keep_mb")) *value = g_tuning.pool_keep_mb;
        else return set_error(EC_ERR_ARG, "ec_stat_get: unknown knob '%s'", k);
    } else if (!std::strcmp(key, "devices")) {
        std::lock_guard<std::mutex> lk(g_mu);
        *value = static_cast<int64_t>(g_devs.size());
    } else if (!std::strcmp(key, "scratch_streams")) {
        std::lock_guard<std::mutex> lk(g_mu);
        int64_t n = 0;
        for (auto& kv : g_devs) n += static_cast<int64_t>(kv.second.scratch.size());
        *value = n;
    } else {
        bool known = false;
        *value = expr_stat(key, &known);  // expr_interp_launches, expr_jit_launches / _compiles / _failures / _programs
        if (!known) return set_error(EC_ERR_ARG, "ec_stat_get: unknown key '%s'", key);
    }
    return EC_OK;
}

// ------------------------------------------------------------------ tuning
extern "C" ec_status ec_tune_set(const char* key, int64_t value) {
    if (!key) return set_error(EC_ERR_ARG, "ec_tune_set: null key");
    if (!std::strcmp(key, "binop_variant")) g_tuning.binop_variant = static_cast<int>(value);
    else if (!std::strcmp(key, "reduce_bpc")) g_tuning.reduce_bpc = value > 0 ? static_cast<int>(value) : 0;
    else if (!std::strcmp(key, "reduce_shape")) g_tuning.reduce_shape = static_cast<int>(value);
    else if (!std::strcmp(key, "map_u")) g_tuning.map_u = static_cast<int>(value);
    else if (!std::strcmp(key, "peel")) g_tuning.peel = static_cast<int>(value);
    else if (!std::strcmp(key, "unaligned_vector")) g_tuning.unaligned_vector = value != 0;
    else if (!std::strcmp(key, "fused_mixed")) g_tuning.fused_mixed = static_cast<int>(value);
    else if (!std::strcmp(key, "mall_mb")) g_tuning.mall_mb = value < 0 ? 0 : value;
    else if (!std::strcmp(key, "inject_shard_failure")) g_tuning.inject_shard_failure = static_cast<int>(value);
    else if (!std::strcmp(key, "inject_pin_refusal")) g_tuning.inject_pin_refusal = value != 0;
    else if (!std::strcmp(key, "expr_fixed")) g_tuning.expr_fixed = value != 0;
    else if (!std::strcmp(key, "write_lds_kb")) g_tuning.write_lds_kb = value < 0 ? 0 : value > 64 ? 64 : static_cast<int>(value);
    else if (!std::strcmp(key, "fused_lds_kb")) g_tuning.fused_lds_kb = value < 0 ? 0 : value > 64 ? 64 : static_cast<int>(value);
    else if (!std::strcmp(key, "binop_lds_kb")) g_tuning.binop_lds_kb = value < 0 ? -1 : value > 64 ? 64 : static_cast<int>(value);
    else if (!std::strcmp(key, "scalar_lds_kb")) g_tuning.scalar_lds_kb = value < 0 ? -1 : value > 64 ? 64 : static_cast<int>(value);
    else if (!std::strcmp(key, "map_lds_kb")) g_tuning.map_lds_kb = value < 0 ? 0 : value > 64 ? 64 : static_cast<int>(value);
    else if (!std::strcmp(key, "counts_one_launch")) g_tuning.counts_one_launch = value < 0 ? 0 : value > 2 ? 2 : static_cast<int>(value);
    else if (!std::strcmp(key, "cache_force")) g_tuning.cache_force = value < 0 ? -1 : static_cast<int>(value);
    else if (!std::strcmp(key, "expr_jit")) g_tuning.expr_jit = value < 0 ? 0 : value > 2 ? 2 : static_cast<int>(value);
    else if (!std::strcmp(key, "pool_keep_mb")) {
        g_tuning.pool_keep_mb = value < 0 ? 0 : value;
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_devs)
            if (kv.second.pool) set_pool_threshold(kv.second.pool);
    } else return set_error(EC_ERR_ARG, "ec_tune_set: unknown key '%s'", key);
    return EC_OK;
}

// ec_sharded.hip — one process driving the n GPUs of a node (SURVEY §8b `ec_shard_*`, §8e).
//
// A raster is cut into contiguous row-blocks, shard i on device i of the group.  Each device has
//   * a launch thread bound to it for life (HIP's current device is per host thread, and a launch costs the
//     host ≈5 µs: eight devices served by one thread would serialise 40 µs of launches in front of kernels
//     that take ≈55 µs per 1/8 shard of a 16384² raster; eight threads issue them side by side),
//   * a non-blocking stream with its reduction scratch,
//   * a 32-byte device payload slot and a 32-byte pinned host slot,
//   * an RCCL communicator of the n-device clique (ncclCommInitAll) unless EC_GROUP_HOST_COMBINE.
//
// Element-wise entry points are FIRE-AND-FORGET (round 3): the calling thread checks the arguments, copies the
// per-shard pointers into one job per launch thread, posts the jobs and returns — it does not wait for the launch
// threads to have issued.  A launch thread that has just run a job polls its queue for a few tens of microseconds
// before it sleeps, so back-to-back sharded calls cost the caller a queue push per device and no futex wake
// (profiles/r03/group_fanout.md).  A failure inside a posted job (a launch error) is recorded in the group and
// returned by the next ec_shard_group_sync or reduction.  Rounds 1-2 blocked every call on a latch until all
// threads had issued (EC_GROUP_BLOCKING_ISSUE keeps that form for comparison).
//
// The reductions are synchronous and run in phases, so that a shard that fails cannot leave the others waiting in a
// collective: (0) every shard's arguments are checked on the calling thread, (1) every shard reduces locally to its
// payload slot — all statuses are collected, (2) only if ALL are EC_OK does every launch thread enqueue the
// ncclAllReduce of its own communicator (the one-thread-per-device use of RCCL needs no group call), (3) copy back and
// wait.  If an enqueue of phase 2 fails on some shard after others have enqueued, the group is poisoned: its
// communicators are aborted (ncclCommAbort) so that no stream waits for the missing rank, and every later call returns
// EC_ERR_RCCL.  Every wave of every kernel launched here finishes on its own (no persistent kernels), so destroying a
// group only has to wait for its streams.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ec_collective.hpp"
#include "ec_hostpipe.hpp"
#include "ec_lattice.hpp"
#include "ec_runtime.hpp"

using namespace ecd;

namespace {

// How long a launch thread polls its queue after a job before it goes to sleep on the condition variable.
constexpr auto kWorkerSpin = std::chrono::microseconds(60);

struct Worker {
    int device = -1;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::atomic<int> pending{0};      // jobs in q (read by the polling worker without the lock)
    std::atomic<bool> sleeping{false};
    bool stop = false;

    void run() {
        (void)ec_set_device(device);  // for the life of the thread
        for (;;) {
            std::function<void()> job;
            if (pending.load(std::memory_order_acquire) == 0) {  // poll, then sleep
                const auto until = std::chrono::steady_clock::now() + kWorkerSpin;
                while (pending.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                if (q.empty()) {
                    sleeping.store(true, std::memory_order_release);
                    cv.wait(lk, [&] { return stop || !q.empty(); });
                    sleeping.store(false, std::memory_order_release);
                    if (q.empty()) return;  // stop requested and drained
                }
                job = std::move(q.front());
                q.pop_front();
                pending.fetch_sub(1, std::memory_order_acq_rel);
            }
            job();
        }
    }
    void post(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(job));
            pending.fetch_add(1, std::memory_order_acq_rel);
        }
        // a polling worker sees `pending`; only a sleeping one needs the wake-up.  (The worker sets `sleeping` under
        // `mu` after finding the queue empty, so a job pushed before that is found, one pushed after sees the flag.)
        if (sleeping.load(std::memory_order_acquire)) cv.notify_one();
    }
};

struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    int left;
    explicit Latch(int n) : left(n) {}
    void arrive() {
        std::lock_guard<std::mutex> lk(mu);
        if (--left == 0) cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return left == 0; });
    }
};

}  // namespace

struct ec_shard_group {
    int n = 0;
    uint32_t flags = 0;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<int64_t*> payload_dev;   // 4 words per shard
    std::vector<int64_t*> payload_host;  // 4 words per shard, pinned
    std::vector<ncclComm_t> comms;       // empty with EC_GROUP_HOST_COMBINE
    std::vector<Worker*> workers;        // empty when n == 1: the caller's thread does the work
    std::mutex call_mu;                  // one sharded call is POSTED at a time per group (keeps every device's queue in
                                         // call order); the synchronous calls hold it to their end (payload slots are per group)
    std::mutex err_mu;                   // first failure of a fire-and-forget job, until a sync / reduction returns it
    ec_status deferred = EC_OK;
    std::string deferred_text;
    std::atomic<bool> poisoned{false};   // a collective was left incomplete: communicators aborted, the group is unusable
    std::atomic<int64_t> posted{0};      // fire-and-forget jobs posted / run (statistics for tools/group_bench.c)
};

namespace {

using ShardFn = std::function<ec_status(int)>;

void record_deferred(ec_shard_group* g, int i, ec_status st, const std::string& text) {
    std::lock_guard<std::mutex> lk(g->err_mu);
    if (g->deferred != EC_OK) return;  // first failure wins
    g->deferred = st;
    g->deferred_text = "shard " + std::to_string(i) + " (device " + std::to_string(g->devices[i]) + "): " + text;
}

ec_status take_deferred(ec_shard_group* g) {
    std::lock_guard<std::mutex> lk(g->err_mu);
    if (g->deferred == EC_OK) return EC_OK;
    const ec_status st = g->deferred;
    g->deferred = EC_OK;
    return set_error_text(st, g->deferred_text + " (reported by a later call: the failing call had already returned)");
}

// fn(0) on the caller's thread with the group's device bound (groups of one shard have no launch threads)
ec_status run_inline(ec_shard_group* g, const ShardFn& fn) {
    int32_t before = -1;
    const bool had = ec_get_device(&before) == EC_OK;
    ec_status st = ec_set_device(g->devices[0]);
    if (st == EC_OK) st = fn(0);
    if (had && before != g->devices[0]) {
        const std::string keep = last_error_text();
        (void)ec_set_device(before);
        if (st != EC_OK) set_error_text(st, keep);
    }
    return st;
}

// fn(i) on every shard's launch thread, concurrently; waits for all; every status is collected, the first failing one
// (and its message) is returned.
ec_status for_each_shard(ec_shard_group* g, const ShardFn& fn, std::vector<ec_status>* all = nullptr) {
    if (g->workers.empty()) {  // n == 1
        const ec_status st = run_inline(g, fn);
        if (all) all->assign(1, st);
        return st;
    }
    std::vector<ec_status> status(g->n, EC_OK);
    std::vector<std::string> text(g->n);
    Latch done(g->n);
    for (int i = 0; i < g->n; ++i) {
        g->workers[i]->post([&, i] {
            status[i] = fn(i);
            if (status[i] != EC_OK) text[i] = last_error_text();  // error text is thread-local: carry it across
            done.arrive();
        });
    }
    done.wait();
    if (all) *all = status;
    for (int i = 0; i < g->n; ++i)
        if (status[i] != EC_OK) return set_error_text(status[i], "shard " + std::to_string(i) + " (device " + std::to_string(g->devices[i]) + "): " + text[i]);
    return EC_OK;
}

// Fire-and-forget: fn owns copies of everything it needs.  Returns once the jobs are posted; failures are deferred.
ec_status post_each_shard(ec_shard_group* g, ShardFn fn) {
    if (g->flags & EC_GROUP_BLOCKING_ISSUE) return for_each_shard(g, fn);
    if (g->workers.empty()) return run_inline(g, fn);  // one shard: issuing IS the call, nothing to hand over
    auto shared = std::make_shared<ShardFn>(std::move(fn));
    for (int i = 0; i < g->n; ++i) {
        g->workers[i]->post([g, i, shared] {
            int want = i + 1;  // test hook (ec_tune_set("inject_shard_failure", shard + 1)): this job fails instead of launching
            const bool injected = tuning().inject_shard_failure.compare_exchange_strong(want, 0);
            const ec_status st = injected ? set_error(EC_ERR_HIP, "injected failure (inject_shard_failure)") : (*shared)(i);
            if (st != EC_OK) record_deferred(g, i, st, last_error_text());
        });
    }
    g->posted.fetch_add(g->n, std::memory_order_relaxed);
    return EC_OK;
}

ec_status check_group(const ec_shard_group* g, const char* what) {
    if (!g || g->n < 1) return set_error(EC_ERR_ARG, "%s: null shard group", what);
    if (g->poisoned.load(std::memory_order_acquire))
        return set_error(EC_ERR_RCCL, "%s: the shard group is poisoned (a collective was left incomplete; destroy the group)", what);
    return EC_OK;
}

// n[i] cells at p[i] for every shard: no null pointer where there are cells (checked on the calling thread, before
// any fan-out: a shard that failed alone would otherwise leave the others inside a collective)
ec_status check_shard_ptrs(const ec_shard_group* g, const char* what, const char* name, const void* const* p, const size_t* n) {
    for (int i = 0; i < g->n; ++i)
        if (n[i] > 0 && !p[i]) return set_error(EC_ERR_ARG, "%s: %s[%d] is null with n[%d] = %zu", what, name, i, i, n[i]);
    return EC_OK;
}

template <typename P>
std::vector<P> copy_n(const P* a, int n) { return std::vector<P>(a, a + n); }

void poison(ec_shard_group* g) {
    g->poisoned.store(true, std::memory_order_release);
    const Rccl* R = rccl("ec_shard_group(poison)");
    for (ncclComm_t c : g->comms)
        if (c && R && R->comm_abort) (void)R->comm_abort(c);  // unblocks the ranks that did enqueue
    g->comms.assign(g->comms.size(), nullptr);
}

// The exchange step of a reduction whose per-shard payloads {a, b} are in payload_dev[i][0..1] on EVERY shard
// (phase 1 succeeded everywhere).  Phase 2: every launch thread enqueues its all-reduce; phase 3: copy back, wait.
ec_status exchange(ec_shard_group* g, bool is_keys) {
    if (!g->comms.empty()) {
        std::vector<ec_status> all;
        ec_status st = for_each_shard(g, [&](int i) {
            return is_keys ? ec_allreduce_min_max_keys(g->comms[i], g->payload_dev[i], g->streams[i])
                           : ec_allreduce_counts(g->comms[i], reinterpret_cast<uint64_t*>(g->payload_dev[i]), g->streams[i]);
        }, &all);
        if (st != EC_OK) {
            bool some_enqueued = false;
            for (ec_status s : all) some_enqueued = some_enqueued || s == EC_OK;
            if (some_enqueued && g->n > 1) {
                const std::string keep = last_error_text();
                poison(g);
                return set_error_text(st, keep + " — other shards had enqueued the collective: group poisoned, communicators aborted");
            }
            return st;
        }
    }
    return for_each_shard(g, [&](int i) {
        ec_status st = check_hip(hipMemcpyAsync(g->payload_host[i], g->payload_dev[i], 2 * sizeof(int64_t), hipMemcpyDeviceToHost, g->streams[i]),
                                 "hipMemcpyAsync(payload)");
        if (st != EC_OK) return st;
        return check_hip(hipStreamSynchronize(g->streams[i]), "hipStreamSynchronize");
    });
}

}  // namespace

extern "C" ec_status ec_shard_group_create(const int32_t* devices, int32_t n, uint32_t flags, ec_shard_group** out) {
    if (!devices || !out || n < 1 || n > 64) return set_error(EC_ERR_ARG, "ec_shard_group_create: null argument or n outside 1..64");
    *out = nullptr;
    const bool host_combine = (flags & EC_GROUP_HOST_COMBINE) != 0;
    if (!host_combine)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (devices[i] == devices[j])
                    return set_error(EC_ERR_ARG, "ec_shard_group_create: device %d listed twice (RCCL needs one rank per GPU; "
                                                 "EC_GROUP_HOST_COMBINE allows it)", int(devices[i]));
    int32_t before = -1;
    const bool had = ec_get_device(&before) == EC_OK;
    std::unique_ptr<ec_shard_group> g(new ec_shard_group);
    g->n = n;
    g->flags = flags;
    g->devices.assign(devices, devices + n);
    g->streams.assign(n, nullptr);
    g->payload_dev.assign(n, nullptr);
    g->payload_host.assign(n, nullptr);
    ec_status st = EC_OK;
    for (int i = 0; i < n && st == EC_OK; ++i) {
        st = ec_init(devices[i]);
        ec_stream s = nullptr;
        if (st == EC_OK) st = ec_stream_create(&s);  // also prepares the stream's reduction scratch
        g->streams[i] = static_cast<hipStream_t>(s);
        if (st == EC_OK) st = check_hip(hipMalloc(reinterpret_cast<void**>(&g->payload_dev[i]), 4 * sizeof(int64_t)), "hipMalloc(payload)");
        if (st == EC_OK) st = check_hip(hipHostMalloc(reinterpret_cast<void**>(&g->payload_host[i]), 4 * sizeof(int64_t), hipHostMallocDefault),
                                        "hipHostMalloc(payload)");
    }
    if (st == EC_OK && !host_combine) {
        std::vector<ec_comm> cs(n, nullptr);
        st = ec_comm_init_all(devices, n, cs.data());
        if (st == EC_OK) {
            g->comms.resize(n);
            for (int i = 0; i < n; ++i) g->comms[i] = static_cast<ncclComm_t>(cs[i]);
        }
    }
    if (st == EC_OK && n > 1) {
        for (int i = 0; i < n; ++i) {
            Worker* w = new Worker;
            w->device = devices[i];
            g->workers.push_back(w);
            w->th = std::thread([w] { w->run(); });
        }
    }
    const std::string keep = last_error_text();
    if (had) (void)ec_set_device(before);
    if (st != EC_OK) {
        (void)ec_shard_group_destroy(g.release());
        return set_error_text(st, keep);
    }
    *out = g.release();
    return EC_OK;
}

extern "C" ec_status ec_shard_group_destroy(ec_shard_group* g) {
    if (!g) return EC_OK;
    for (Worker* w : g->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->stop = true;
        }
        w->cv.notify_one();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    g->workers.clear();
    int32_t before = -1;
    const bool had = ec_get_device(&before) == EC_OK;
    for (int i = 0; i < g->n; ++i) {
        if (ec_set_device(g->devices[i]) != EC_OK) continue;
        if (g->streams[i]) (void)hipStreamSynchronize(g->streams[i]);
        if (i < int(g->comms.size()) && g->comms[i]) (void)ec_comm_destroy(g->comms[i]);
        if (g->payload_dev[i]) (void)hipFree(g->payload_dev[i]);
        if (g->payload_host[i]) (void)hipHostFree(g->payload_host[i]);
        if (g->streams[i]) (void)ec_stream_destroy(g->streams[i]);
    }
    if (had) (void)ec_set_device(before);
    delete g;
    return EC_OK;
}

extern "C" int32_t ec_shard_group_size(const ec_shard_group* g) { return g ? g->n : 0; }

extern "C" ec_status ec_shard_group_shard(const ec_shard_group* g, int32_t shard, int32_t* device, ec_stream* stream) {
    ec_status st = check_group(g, "ec_shard_group_shard");
    if (st != EC_OK) return st;
    if (shard < 0 || shard >= g->n) return set_error(EC_ERR_ARG, "ec_shard_group_shard: shard %d of %d", int(shard), g->n);
    if (device) *device = g->devices[shard];
    if (stream) *stream = g->streams[shard];
    return EC_OK;
}

extern "C" ec_status ec_shard_group_foreach(ec_shard_group* g, ec_shard_fn fn, void* user) {
    ec_status st = check_group(g, "ec_shard_group_foreach");
    if (st != EC_OK) return st;
    if (!fn) return set_error(EC_ERR_ARG, "ec_shard_group_foreach: null callback");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) { return fn(i, g->devices[i], g->streams[i], user); });
}

// Waits until every posted job has been issued and every shard's stream has drained; returns the first failure a
// fire-and-forget call left behind (and clears it), else the streams' own status.
extern "C" ec_status ec_shard_group_sync(ec_shard_group* g) {
    if (!g || g->n < 1) return set_error(EC_ERR_ARG, "ec_shard_group_sync: null shard group");
    std::lock_guard<std::mutex> lk(g->call_mu);
    const ec_status st = for_each_shard(g, [&](int i) { return check_hip(hipStreamSynchronize(g->streams[i]), "hipStreamSynchronize"); });
    const std::string keep = last_error_text();
    const ec_status deferred = take_deferred(g);
    if (deferred != EC_OK) return deferred;
    if (st != EC_OK) return set_error_text(st, keep);
    return check_group(g, "ec_shard_group_sync");  // a poisoned group says so
}

extern "C" ec_status ec_shard_group_stat(const ec_shard_group* g, const char* key, int64_t* value) {
    if (!g || !key || !value) return set_error(EC_ERR_ARG, "ec_shard_group_stat: null argument");
    if (!std::strcmp(key, "jobs_posted")) *value = g->posted.load(std::memory_order_relaxed);
    else if (!std::strcmp(key, "poisoned")) *value = g->poisoned.load(std::memory_order_acquire) ? 1 : 0;
    else if (!std::strcmp(key, "blocking_issue")) *value = (g->flags & EC_GROUP_BLOCKING_ISSUE) ? 1 : 0;
    else return set_error(EC_ERR_ARG, "ec_shard_group_stat: unknown key '%s'", key);
    return EC_OK;
}

extern "C" ec_status ec_sharded_alloc(ec_shard_group* g, const size_t* bytes, void** dptrs) {
    ec_status st = check_group(g, "ec_sharded_alloc");
    if (st != EC_OK) return st;
    if (!bytes || !dptrs) return set_error(EC_ERR_ARG, "ec_sharded_alloc: null argument");
    for (int i = 0; i < g->n; ++i) dptrs[i] = nullptr;
    std::lock_guard<std::mutex> lk(g->call_mu);
    st = for_each_shard(g, [&](int i) { return ec_alloc(&dptrs[i], bytes[i]); });
    if (st != EC_OK) {
        const std::string keep = last_error_text();
        (void)for_each_shard(g, [&](int i) { ec_status f = ec_free(dptrs[i]); dptrs[i] = nullptr; return f; });
        return set_error_text(st, keep);
    }
    return EC_OK;
}

// (queued behind the launches posted so far on each device; hipFree then waits for the device)
extern "C" ec_status ec_sharded_free(ec_shard_group* g, void* const* dptrs) {
    if (!g || g->n < 1) return set_error(EC_ERR_ARG, "ec_sharded_free: null shard group");  // a poisoned group still frees
    if (!dptrs) return set_error(EC_ERR_ARG, "ec_sharded_free: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) { return ec_free(dptrs[i]); });
}

extern "C" ec_status ec_sharded_upload(ec_shard_group* g, void* const* dst_dev, const void* src_host,
                                       const size_t* byte_offsets, const size_t* bytes) {
    ec_status st = check_group(g, "ec_sharded_upload");
    if (st != EC_OK) return st;
    if (!dst_dev || !byte_offsets || !bytes) return set_error(EC_ERR_ARG, "ec_sharded_upload: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) {
        return ec_upload(dst_dev[i], static_cast<const char*>(src_host) + byte_offsets[i], bytes[i], g->streams[i]);
    });
}

extern "C" ec_status ec_sharded_download(ec_shard_group* g, void* dst_host, const void* const* src_dev,
                                         const size_t* byte_offsets, const size_t* bytes) {
    ec_status st = check_group(g, "ec_sharded_download");
    if (st != EC_OK) return st;
    if (!src_dev || !byte_offsets || !bytes) return set_error(EC_ERR_ARG, "ec_sharded_download: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) {
        return ec_download(static_cast<char*>(dst_host) + byte_offsets[i], src_dev[i], bytes[i], g->streams[i]);
    });
}

// ---- element-wise: checked on the calling thread, posted, not waited for
extern "C" ec_status ec_sharded_binop(ec_shard_group* g, ec_op op, ec_dtype lt, const void* const* l, ec_dtype rt,
                                      const void* const* r, const size_t* n, double* const* out) {
    ec_status st = check_group(g, "ec_sharded_binop");
    if (st != EC_OK) return st;
    if (!l || !r || !n || !out) return set_error(EC_ERR_ARG, "ec_sharded_binop: null argument");
    if (op < EC_ADD || op > EC_DIV) return set_error(EC_ERR_ARG, "ec_sharded_binop: bad op");
    if (!ecl::valid(lt) || !ecl::valid(rt)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_sharded_binop: bad dtype");
    if ((st = check_shard_ptrs(g, "ec_sharded_binop", "l", l, n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, "ec_sharded_binop", "r", r, n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, "ec_sharded_binop", "out", reinterpret_cast<const void* const*>(out), n)) != EC_OK) return st;
    std::lock_guard<std::mutex> lk(g->call_mu);
    return post_each_shard(g, [g, op, lt, rt, l = copy_n(l, g->n), r = copy_n(r, g->n), n = copy_n(n, g->n), out = copy_n(out, g->n)](int i) {
        return ec_binop(op, lt, l[i], rt, r[i], n[i], out[i], g->streams[i]);
    });
}

extern "C" ec_status ec_sharded_masked_binop(ec_shard_group* g, ec_op op, ec_dtype lt, const void* const* l, const uint8_t* const* lmask,
                                             ec_dtype rt, const void* const* r, const uint8_t* const* rmask, const size_t* n,
                                             double* const* out, uint8_t* const* out_mask) {
    ec_status st = check_group(g, "ec_sharded_masked_binop");
    if (st != EC_OK) return st;
    if (!l || !lmask || !r || !rmask || !n || !out || !out_mask) return set_error(EC_ERR_ARG, "ec_sharded_masked_binop: null argument");
    if (op < EC_ADD || op > EC_DIV) return set_error(EC_ERR_ARG, "ec_sharded_masked_binop: bad op");
    if (!ecl::valid(lt) || !ecl::valid(rt)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_sharded_masked_binop: bad dtype");
    const char* what = "ec_sharded_masked_binop";
    if ((st = check_shard_ptrs(g, what, "l", l, n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, what, "r", r, n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, what, "lmask", reinterpret_cast<const void* const*>(lmask), n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, what, "rmask", reinterpret_cast<const void* const*>(rmask), n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, what, "out", reinterpret_cast<const void* const*>(out), n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, what, "out_mask", reinterpret_cast<const void* const*>(out_mask), n)) != EC_OK) return st;
    std::lock_guard<std::mutex> lk(g->call_mu);
    return post_each_shard(g, [g, op, lt, rt, l = copy_n(l, g->n), lm = copy_n(lmask, g->n), r = copy_n(r, g->n), rm = copy_n(rmask, g->n),
                               n = copy_n(n, g->n), out = copy_n(out, g->n), om = copy_n(out_mask, g->n)](int i) {
        return ec_masked_binop(op, lt, l[i], lm[i], rt, r[i], rm[i], n[i], out[i], om[i], g->streams[i]);
    });
}

extern "C" ec_status ec_sharded_convert(ec_shard_group* g, ec_dtype st_, const void* const* src, ec_dtype dt, void* const* dst,
                                        const size_t* n) {
    ec_status st = check_group(g, "ec_sharded_convert");
    if (st != EC_OK) return st;
    if (!ecl::valid(st_) || !ecl::valid(dt)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_sharded_convert: bad dtype");
    if (!ec_can_fit_into(st_, dt)) return ec_convert(st_, nullptr, dt, nullptr, 0, nullptr);  // EC_ERR_NARROWING, before any device work
    if (!src || !dst || !n) return set_error(EC_ERR_ARG, "ec_sharded_convert: null argument");
    if ((st = check_shard_ptrs(g, "ec_sharded_convert", "src", src, n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, "ec_sharded_convert", "dst", dst, n)) != EC_OK) return st;
    std::lock_guard<std::mutex> lk(g->call_mu);
    return post_each_shard(g, [g, st_, dt, src = copy_n(src, g->n), dst = copy_n(dst, g->n), n = copy_n(n, g->n)](int i) {
        return ec_convert(st_, src[i], dt, dst[i], n[i], g->streams[i]);
    });
}

extern "C" ec_status ec_sharded_mask_from_nodata(ec_shard_group* g, ec_dtype t, const void* const* p, const size_t* n,
                                                 const ec_value* nd_or_null, uint8_t* const* mask) {
    ec_status st = check_group(g, "ec_sharded_mask_from_nodata");
    if (st != EC_OK) return st;
    if (!p || !n || !mask) return set_error(EC_ERR_ARG, "ec_sharded_mask_from_nodata: null argument");
    if (!ecl::valid(t)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_sharded_mask_from_nodata: bad dtype");
    if ((st = check_shard_ptrs(g, "ec_sharded_mask_from_nodata", "p", p, n)) != EC_OK) return st;
    if ((st = check_shard_ptrs(g, "ec_sharded_mask_from_nodata", "mask", reinterpret_cast<const void* const*>(mask), n)) != EC_OK) return st;
    const bool has_nd = nd_or_null != nullptr;
    const ec_value nd = has_nd ? *nd_or_null : ec_value{};
    std::lock_guard<std::mutex> lk(g->call_mu);
    return post_each_shard(g, [g, t, has_nd, nd, p = copy_n(p, g->n), n = copy_n(n, g->n), mask = copy_n(mask, g->n)](int i) {
        return ec_mask_from_nodata(t, p[i], n[i], has_nd ? &nd : nullptr, mask[i], g->streams[i]);
    });
}

extern "C" ec_status ec_sharded_fused(ec_shard_group* g, ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void* const* const p[4],
                                      const uint8_t* const* const masks_or_null[4], const ec_value* scalars_or_null, const size_t* n,
                                      double* const* out, uint8_t* const* out_mask_or_null) {
    ec_status st = check_group(g, "ec_sharded_fused");
    if (st != EC_OK) return st;
    if (!dt || !p || !n || !out) return set_error(EC_ERR_ARG, "ec_sharded_fused: null argument");
    if ((masks_or_null != nullptr) != (out_mask_or_null != nullptr))
        return set_error(EC_ERR_ARG, "ec_sharded_fused: masks and out_mask go together");
    const int nops = o3 == static_cast<ec_op>(-1) ? 3 : 4;
    struct Call {
        ec_op o1, o2, o3;
        ec_dtype dt[4];
        bool has_p[4], has_m[4], masked, has_sc;
        std::vector<const void*> p[4];
        std::vector<const uint8_t*> m[4];
        ec_value sc[4];
        std::vector<size_t> n;
        std::vector<double*> out;
        std::vector<uint8_t*> om;
    };
    auto c = std::make_shared<Call>();
    c->o1 = o1; c->o2 = o2; c->o3 = o3;
    c->masked = masks_or_null != nullptr;
    c->has_sc = scalars_or_null != nullptr;
    for (int k = 0; k < 4; ++k) {
        c->dt[k] = dt[k];
        c->has_p[k] = k < nops && p[k] != nullptr;
        c->has_m[k] = c->has_p[k] && masks_or_null && masks_or_null[k] != nullptr;
        if (c->has_p[k]) {
            if ((st = check_shard_ptrs(g, "ec_sharded_fused", "p[k]", p[k], n)) != EC_OK) return st;
            c->p[k] = copy_n(p[k], g->n);
        }
        if (c->has_m[k]) c->m[k] = copy_n(masks_or_null[k], g->n);
        c->sc[k] = scalars_or_null ? scalars_or_null[k] : ec_value{};
    }
    if ((st = check_shard_ptrs(g, "ec_sharded_fused", "out", reinterpret_cast<const void* const*>(out), n)) != EC_OK) return st;
    c->n = copy_n(n, g->n);
    c->out = copy_n(out, g->n);
    if (out_mask_or_null) c->om = copy_n(out_mask_or_null, g->n);
    std::lock_guard<std::mutex> lk(g->call_mu);
    return post_each_shard(g, [g, c](int i) {
        const void* pi[4];
        const uint8_t* mi[4];
        for (int k = 0; k < 4; ++k) {
            pi[k] = c->has_p[k] ? c->p[k][i] : nullptr;  // no pointer array: operand k is the scalar scalars[k]
            mi[k] = c->has_m[k] ? c->m[k][i] : nullptr;
        }
        const ec_value* sc = c->has_sc ? c->sc : nullptr;
        if (c->masked) return ec_masked_fused(c->o1, c->o2, c->o3, c->dt, pi, mi, sc, c->n[i], c->out[i], c->om[i], g->streams[i]);
        return ec_fused(c->o1, c->o2, c->o3, c->dt, pi, sc, c->n[i], c->out[i], g->streams[i]);
    });
}

// Expression programs (ec_expr / ec_masked_expr) on every shard.  p[k] / masks_or_null[k]: stream k's per-shard pointers.
// Everything that can be refused is refused here, on the calling thread, the program included.
extern "C" ec_status ec_sharded_expr(ec_shard_group* g, const ec_dtype* dt, const void* const* const* p,
                                     const uint8_t* const* const* masks_or_null, int32_t n_streams, const ec_value* scalars,
                                     int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, const size_t* n, double* const* out,
                                     uint8_t* const* out_mask_or_null) {
    ec_status st = check_group(g, "ec_sharded_expr");
    if (st != EC_OK) return st;
    if (!dt || !p || !steps || !n || !out || (n_scalars > 0 && !scalars)) return set_error(EC_ERR_ARG, "ec_sharded_expr: null argument");
    if (n_streams < 1 || n_streams > 4 || n_scalars < 0 || n_scalars > 8 || n_steps < 1 || n_steps > 16)
        return set_error(EC_ERR_ARG, "ec_sharded_expr: %d streams (1..4), %d scalars (0..8), %d steps (1..16)", int(n_streams), int(n_scalars), int(n_steps));
    if ((masks_or_null != nullptr) != (out_mask_or_null != nullptr)) return set_error(EC_ERR_ARG, "ec_sharded_expr: masks and out_mask go together");
    {   // the program: parsed once here (no device needed), so that a malformed one is refused before anything is posted
        size_t len = 0;
        if ((st = ec_expr_source(dt, n_streams, n_scalars, steps, n_steps, nullptr, nullptr, 0, &len)) != EC_OK) return st;
    }
    struct Call {
        int32_t ns, nsc, nst;
        ec_dtype dt[4];
        ec_value sc[8];
        ec_expr_step steps[16];
        bool masked;
        std::vector<const void*> p[4];
        std::vector<const uint8_t*> m[4];
        std::vector<size_t> n;
        std::vector<double*> out;
        std::vector<uint8_t*> om;
    };
    auto c = std::make_shared<Call>();
    c->ns = n_streams; c->nsc = n_scalars; c->nst = n_steps;
    c->masked = masks_or_null != nullptr;
    for (int k = 0; k < n_streams; ++k) {
        c->dt[k] = dt[k];
        if (!p[k] || (c->masked && !masks_or_null[k])) return set_error(EC_ERR_ARG, "ec_sharded_expr: stream %d has no pointer array", k);
        if ((st = check_shard_ptrs(g, "ec_sharded_expr", "p[k]", p[k], n)) != EC_OK) return st;
        c->p[k] = copy_n(p[k], g->n);
        if (c->masked) {
            if ((st = check_shard_ptrs(g, "ec_sharded_expr", "masks[k]", reinterpret_cast<const void* const*>(masks_or_null[k]), n)) != EC_OK) return st;
            c->m[k] = copy_n(masks_or_null[k], g->n);
        }
    }
    for (int k = 0; k < n_scalars; ++k) c->sc[k] = scalars[k];
    for (int k = 0; k < n_steps; ++k) c->steps[k] = steps[k];
    if ((st = check_shard_ptrs(g, "ec_sharded_expr", "out", reinterpret_cast<const void* const*>(out), n)) != EC_OK) return st;
    c->n = copy_n(n, g->n);
    c->out = copy_n(out, g->n);
    if (out_mask_or_null) {
        if ((st = check_shard_ptrs(g, "ec_sharded_expr", "out_mask", reinterpret_cast<const void* const*>(out_mask_or_null), n)) != EC_OK) return st;
        c->om = copy_n(out_mask_or_null, g->n);
    }
    std::lock_guard<std::mutex> lk(g->call_mu);
    return post_each_shard(g, [g, c](int i) {
        const void* pi[4] = {nullptr, nullptr, nullptr, nullptr};
        const uint8_t* mi[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < c->ns; ++k) {
            pi[k] = c->p[k][i];
            if (c->masked) mi[k] = c->m[k][i];
        }
        if (c->masked) return ec_masked_expr(c->dt, pi, mi, c->ns, c->sc, c->nsc, c->steps, c->nst, c->n[i], c->out[i], c->om[i], g->streams[i]);
        return ec_expr(c->dt, pi, c->ns, c->sc, c->nsc, c->steps, c->nst, c->n[i], c->out[i], g->streams[i]);
    });
}

// Host memory in, host memory out over ALL the GPUs of the group: the raster's row-blocks (ec_shard_range) go through
// ec_host_expr / ec_host_masked_expr side by side, one pipeline per launch thread — every GPU has its own PCIe link, so a
// host-resident raster is bound by the sum of the links (and then by host memory), not by one of them.  Synchronous.
extern "C" ec_status ec_sharded_host_expr(ec_shard_group* g, const ec_dtype* dt, const void* const* p_host, const ec_value* const* nodata_or_null,
                                          int32_t n_streams, const ec_value* scalars, int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps,
                                          uint64_t n_rows, uint64_t n_cols, double* out_host, const double* out_nodata_or_null,
                                          uint8_t* out_mask_host_or_null, size_t chunk_cells) {
    ec_status st = check_group(g, "ec_sharded_host_expr");
    if (st != EC_OK) return st;
    if (!dt || !p_host || !steps || !out_host) return set_error(EC_ERR_ARG, "ec_sharded_host_expr: null argument");
    {   // the program, once, on the calling thread
        size_t len = 0;
        if ((st = ec_expr_source(dt, n_streams, n_scalars, steps, n_steps, nullptr, nullptr, 0, &len)) != EC_OK) return st;
    }
    for (int k = 0; k < n_streams; ++k)
        if (!p_host[k]) return set_error(EC_ERR_ARG, "ec_sharded_host_expr: stream %d is null", k);
    // everything a shard's pipeline would refuse is refused here, on the calling thread, before a page is locked
    if (nodata_or_null)
        for (int k = 0; k < n_streams; ++k)
            if (nodata_or_null[k] && nodata_or_null[k]->dtype != dt[k])
                return set_error(EC_ERR_ARG, "ec_sharded_host_expr: nodata[%d] has cell type %d, its stream %d", k, int(nodata_or_null[k]->dtype), int(dt[k]));
    if (n_cols != 0 && n_rows > (UINT64_MAX / 8) / n_cols)  // n cells of at most 8 bytes: the byte counts below stay inside 64 bits
        return set_error(EC_ERR_ARG, "ec_sharded_host_expr: %llu x %llu cells overflow the address space", (unsigned long long)n_rows, (unsigned long long)n_cols);
    if ((st = ensure_ready()) != EC_OK) return st;
    // the WHOLE arrays are page-locked once, here; the shards' pipelines find their row-blocks inside these registrations
    // (row-blocks end in the middle of pages: registering them one by one would collide on the shared pages).  Arrays the runtime
    // refuses to register stay in the table as ranges in use; the shards share those entries and copy
    // through the pageable path (PinSet, ec_hostpipe.hip)
    const uint64_t n = n_rows * n_cols;
    PinSet whole;
    {
        std::vector<std::pair<const void*, size_t>> ranges;
        for (int k = 0; k < n_streams; ++k) ranges.emplace_back(p_host[k], n * ecl::size_of(dt[k]));
        ranges.emplace_back(out_host, n * sizeof(double));
        if (out_mask_host_or_null) ranges.emplace_back(out_mask_host_or_null, n);
        // small rasters are not registered at all (ec_hostpipe.hip: their pages are shared heap pages the runtime pins itself); the shards'
        // pipelines decide the same way, each for its row-block
        size_t link = 0;
        for (const auto& r : ranges) link += r.second;
        if (link > (size_t(64) << 20)) whole.pin_all(ranges);
        else whole.use_all(ranges);
    }
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) {
        uint64_t off = 0, len = 0;
        ec_status s = ec_shard_range(n_rows, n_cols, static_cast<uint32_t>(i), static_cast<uint32_t>(g->n), &off, &len);
        if (s != EC_OK || len == 0) return s;
        const void* p[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < n_streams; ++k) p[k] = static_cast<const char*>(p_host[k]) + off * ecl::size_of(dt[k]);
        if (nodata_or_null)
            return ec_host_masked_expr(dt, p, nodata_or_null, n_streams, scalars, n_scalars, steps, n_steps, len, out_host + off, out_nodata_or_null,
                                       out_mask_host_or_null ? out_mask_host_or_null + off : nullptr, chunk_cells);
        return ec_host_expr(dt, p, n_streams, scalars, n_scalars, steps, n_steps, len, out_host + off, chunk_cells);
    });
}

// ---- reductions: synchronous, phased (see the head of this file)
extern "C" ec_status ec_sharded_min_max(ec_shard_group* g, ec_dtype t, const void* const* p, const uint8_t* const* masks_or_null,
                                        const size_t* n, ec_value* mn, ec_value* mx) {
    ec_status st = check_group(g, "ec_sharded_min_max");
    if (st != EC_OK) return st;
    if (!p || !n || !mn || !mx) return set_error(EC_ERR_ARG, "ec_sharded_min_max: null argument");
    if (!ecl::valid(t)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_sharded_min_max: bad dtype");
    if ((st = check_shard_ptrs(g, "ec_sharded_min_max", "p", p, n)) != EC_OK) return st;
    if (masks_or_null && (st = check_shard_ptrs(g, "ec_sharded_min_max", "masks", reinterpret_cast<const void* const*>(masks_or_null), n)) != EC_OK) return st;
    std::lock_guard<std::mutex> lk(g->call_mu);
    // phase 1: local reductions everywhere (an empty shard contributes the idempotent sentinels (T::MAX, T::MIN): src/buffer.rs:170)
    st = for_each_shard(g, [&](int i) {
        return ec_min_max_keys(t, p[i], masks_or_null ? masks_or_null[i] : nullptr, n[i], g->payload_dev[i], g->streams[i]);
    });
    const std::string keep = last_error_text();
    const ec_status deferred = take_deferred(g);  // an earlier fire-and-forget call that failed: its output may feed this reduction
    if (deferred != EC_OK) return deferred;
    if (st != EC_OK) return set_error_text(st, keep);  // nothing was enqueued on any communicator
    st = exchange(g, true);
    if (st != EC_OK) return st;
    int64_t keys[2] = {g->payload_host[0][0], g->payload_host[0][1]};
    if (g->comms.empty())  // host combine: element-wise MAX of {~key(min), key(max)}
        for (int i = 1; i < g->n; ++i) {
            if (g->payload_host[i][0] > keys[0]) keys[0] = g->payload_host[i][0];
            if (g->payload_host[i][1] > keys[1]) keys[1] = g->payload_host[i][1];
        }
    return ec_min_max_decode(t, keys, mn, mx);
}

// min_max of an expression program's result over the whole sharded raster, without the raster: ec_expr_min_max_keys per shard
// (the compiled reduce kernel, or two passes until the program is compiled), then the same 16-byte MAX exchange as
// ec_sharded_min_max.  Synchronous, phased like the other reductions.
extern "C" ec_status ec_sharded_expr_min_max(ec_shard_group* g, const ec_dtype* dt, const void* const* const* p,
                                             const uint8_t* const* const* masks_or_null, int32_t n_streams, const ec_value* scalars,
                                             int32_t n_scalars, const ec_expr_step* steps, int32_t n_steps, const size_t* n, ec_value* mn,
                                             ec_value* mx) {
    ec_status st = check_group(g, "ec_sharded_expr_min_max");
    if (st != EC_OK) return st;
    if (!dt || !p || !steps || !n || !mn || !mx || (n_scalars > 0 && !scalars)) return set_error(EC_ERR_ARG, "ec_sharded_expr_min_max: null argument");
    {   // the program, once, on the calling thread
        size_t len = 0;
        if ((st = ec_expr_source(dt, n_streams, n_scalars, steps, n_steps, nullptr, nullptr, 0, &len)) != EC_OK) return st;
    }
    for (int k = 0; k < n_streams; ++k) {
        if (!p[k] || (masks_or_null && !masks_or_null[k])) return set_error(EC_ERR_ARG, "ec_sharded_expr_min_max: stream %d has no pointer array", k);
        if ((st = check_shard_ptrs(g, "ec_sharded_expr_min_max", "p[k]", p[k], n)) != EC_OK) return st;
        if (masks_or_null && (st = check_shard_ptrs(g, "ec_sharded_expr_min_max", "masks[k]", reinterpret_cast<const void* const*>(masks_or_null[k]), n)) != EC_OK) return st;
    }
    std::lock_guard<std::mutex> lk(g->call_mu);
    st = for_each_shard(g, [&](int i) {
        const void* pi[4] = {nullptr, nullptr, nullptr, nullptr};
        const uint8_t* mi[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < n_streams; ++k) {
            pi[k] = p[k][i];
            if (masks_or_null) mi[k] = masks_or_null[k][i];
        }
        return ec_expr_min_max_keys(dt, pi, masks_or_null ? mi : nullptr, n_streams, scalars, n_scalars, steps, n_steps, n[i], g->payload_dev[i], g->streams[i]);
    });
    const std::string keep = last_error_text();
    const ec_status deferred = take_deferred(g);
    if (deferred != EC_OK) return deferred;
    if (st != EC_OK) return set_error_text(st, keep);
    st = exchange(g, true);
    if (st != EC_OK) return st;
    int64_t keys[2] = {g->payload_host[0][0], g->payload_host[0][1]};
    if (g->comms.empty())
        for (int i = 1; i < g->n; ++i) {
            if (g->payload_host[i][0] > keys[0]) keys[0] = g->payload_host[i][0];
            if (g->payload_host[i][1] > keys[1]) keys[1] = g->payload_host[i][1];
        }
    return ec_min_max_decode(EC_F64, keys, mn, mx);
}

extern "C" ec_status ec_sharded_counts(ec_shard_group* g, const uint8_t* const* masks, const size_t* n, uint64_t* n_true,
                                       uint64_t* n_false) {
    ec_status st = check_group(g, "ec_sharded_counts");
    if (st != EC_OK) return st;
    if (!masks || !n || !n_true || !n_false) return set_error(EC_ERR_ARG, "ec_sharded_counts: null argument");
    if ((st = check_shard_ptrs(g, "ec_sharded_counts", "masks", reinterpret_cast<const void* const*>(masks), n)) != EC_OK) return st;
    std::lock_guard<std::mutex> lk(g->call_mu);
    st = for_each_shard(g, [&](int i) {
        return ec_mask_counts_device(masks[i], n[i], reinterpret_cast<uint64_t*>(g->payload_dev[i]), g->streams[i]);
    });
    const std::string keep = last_error_text();
    const ec_status deferred = take_deferred(g);
    if (deferred != EC_OK) return deferred;
    if (st != EC_OK) return set_error_text(st, keep);
    st = exchange(g, false);
    if (st != EC_OK) return st;
    uint64_t a = static_cast<uint64_t>(g->payload_host[0][0]), b = static_cast<uint64_t>(g->payload_host[0][1]);
    if (g->comms.empty())
        for (int i = 1; i < g->n; ++i) {
            a += static_cast<uint64_t>(g->payload_host[i][0]);
            b += static_cast<uint64_t>(g->payload_host[i][1]);
        }
    *n_true = a;
    *n_false = b;
    return EC_OK;
}

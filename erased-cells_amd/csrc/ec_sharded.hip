// ec_sharded.hip — one process driving the n GPUs of a node (SURVEY §8b `ec_shard_*`, §8e).
//
// A raster is cut into contiguous row-blocks, shard i on device i of the group.  Each device has
//   * a launch thread bound to it for life (HIP's current device is per host thread, and a launch costs the
//     host ≈5 µs: eight devices served by one thread would serialise 40 µs of launches in front of kernels
//     that take ≈55 µs per 1/8 shard of a 16384² raster; eight threads issue them side by side),
//   * a non-blocking stream with its reduction scratch,
//   * a 32-byte device payload slot and a 32-byte pinned host slot,
//   * an RCCL communicator of the n-device clique (ncclCommInitAll) unless EC_GROUP_HOST_COMBINE.
// Element-wise entry points fan out and return; the reductions fan out, all-reduce their 16-byte payloads over
// xGMI (each launch thread issues the ncclAllReduce of its own communicator — the one-thread-per-device use of
// RCCL needs no group call), copy the payload back and wait.  Every wave of every kernel launched here finishes
// on its own (no persistent kernels), so destroying a group only has to wait for its streams.
#include <hip/hip_runtime.h>

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
#include "ec_runtime.hpp"

using namespace ecd;

namespace {

struct Worker {
    int device = -1;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;

    void run() {
        (void)ec_set_device(device);  // for the life of the thread
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;  // stop requested and drained
                job = std::move(q.front());
                q.pop_front();
            }
            job();
        }
    }
    void post(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(job));
        }
        cv.notify_one();
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
    std::mutex call_mu;                  // one sharded call at a time per group (payload slots are per group)
};

namespace {

// fn(i) on every shard's launch thread, concurrently; first failing status (and its message) wins.
ec_status for_each_shard(ec_shard_group* g, const std::function<ec_status(int)>& fn) {
    if (g->workers.empty()) {  // n == 1
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
    for (int i = 0; i < g->n; ++i)
        if (status[i] != EC_OK) return set_error_text(status[i], "shard " + std::to_string(i) + " (device " + std::to_string(g->devices[i]) + "): " + text[i]);
    return EC_OK;
}

ec_status check_group(const ec_shard_group* g, const char* what) {
    if (!g || g->n < 1) return set_error(EC_ERR_ARG, "%s: null shard group", what);
    return EC_OK;
}

// After the per-shard payloads {a, b} are in payload_dev[i][0..1]: exchange, bring them to the host, wait.
ec_status exchange(ec_shard_group* g, int i, bool is_keys) {
    hipStream_t s = g->streams[i];
    if (!g->comms.empty()) {
        ec_status st = is_keys ? ec_allreduce_min_max_keys(g->comms[i], g->payload_dev[i], s)
                               : ec_allreduce_counts(g->comms[i], reinterpret_cast<uint64_t*>(g->payload_dev[i]), s);
        if (st != EC_OK) return st;
    }
    ec_status st = check_hip(hipMemcpyAsync(g->payload_host[i], g->payload_dev[i], 2 * sizeof(int64_t), hipMemcpyDeviceToHost, s),
                             "hipMemcpyAsync(payload)");
    if (st != EC_OK) return st;
    return check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
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

extern "C" ec_status ec_shard_group_sync(ec_shard_group* g) {
    ec_status st = check_group(g, "ec_shard_group_sync");
    if (st != EC_OK) return st;
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) { return check_hip(hipStreamSynchronize(g->streams[i]), "hipStreamSynchronize"); });
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

extern "C" ec_status ec_sharded_free(ec_shard_group* g, void* const* dptrs) {
    ec_status st = check_group(g, "ec_sharded_free");
    if (st != EC_OK) return st;
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

extern "C" ec_status ec_sharded_binop(ec_shard_group* g, ec_op op, ec_dtype lt, const void* const* l, ec_dtype rt,
                                      const void* const* r, const size_t* n, double* const* out) {
    ec_status st = check_group(g, "ec_sharded_binop");
    if (st != EC_OK) return st;
    if (!l || !r || !n || !out) return set_error(EC_ERR_ARG, "ec_sharded_binop: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) { return ec_binop(op, lt, l[i], rt, r[i], n[i], out[i], g->streams[i]); });
}

extern "C" ec_status ec_sharded_masked_binop(ec_shard_group* g, ec_op op, ec_dtype lt, const void* const* l, const uint8_t* const* lmask,
                                             ec_dtype rt, const void* const* r, const uint8_t* const* rmask, const size_t* n,
                                             double* const* out, uint8_t* const* out_mask) {
    ec_status st = check_group(g, "ec_sharded_masked_binop");
    if (st != EC_OK) return st;
    if (!l || !lmask || !r || !rmask || !n || !out || !out_mask) return set_error(EC_ERR_ARG, "ec_sharded_masked_binop: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) {
        return ec_masked_binop(op, lt, l[i], lmask[i], rt, r[i], rmask[i], n[i], out[i], out_mask[i], g->streams[i]);
    });
}

extern "C" ec_status ec_sharded_convert(ec_shard_group* g, ec_dtype st_, const void* const* src, ec_dtype dt, void* const* dst,
                                        const size_t* n) {
    ec_status st = check_group(g, "ec_sharded_convert");
    if (st != EC_OK) return st;
    if (!ec_can_fit_into(st_, dt)) return ec_convert(st_, nullptr, dt, nullptr, 0, nullptr);  // EC_ERR_NARROWING, before any device work
    if (!src || !dst || !n) return set_error(EC_ERR_ARG, "ec_sharded_convert: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) { return ec_convert(st_, src[i], dt, dst[i], n[i], g->streams[i]); });
}

extern "C" ec_status ec_sharded_mask_from_nodata(ec_shard_group* g, ec_dtype t, const void* const* p, const size_t* n,
                                                 const ec_value* nd_or_null, uint8_t* const* mask) {
    ec_status st = check_group(g, "ec_sharded_mask_from_nodata");
    if (st != EC_OK) return st;
    if (!p || !n || !mask) return set_error(EC_ERR_ARG, "ec_sharded_mask_from_nodata: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) { return ec_mask_from_nodata(t, p[i], n[i], nd_or_null, mask[i], g->streams[i]); });
}

extern "C" ec_status ec_sharded_fused(ec_shard_group* g, ec_op o1, ec_op o2, ec_op o3, const ec_dtype dt[4], const void* const* const p[4],
                                      const uint8_t* const* const masks_or_null[4], const ec_value* scalars_or_null, const size_t* n,
                                      double* const* out, uint8_t* const* out_mask_or_null) {
    ec_status st = check_group(g, "ec_sharded_fused");
    if (st != EC_OK) return st;
    if (!dt || !p || !n || !out) return set_error(EC_ERR_ARG, "ec_sharded_fused: null argument");
    if ((masks_or_null != nullptr) != (out_mask_or_null != nullptr))
        return set_error(EC_ERR_ARG, "ec_sharded_fused: masks and out_mask go together");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return for_each_shard(g, [&](int i) {
        const void* pi[4];
        const uint8_t* mi[4];
        for (int k = 0; k < 4; ++k) {
            pi[k] = p[k] ? p[k][i] : nullptr;  // p[k] == NULL: operand k is the scalar scalars[k]
            mi[k] = (masks_or_null && masks_or_null[k]) ? masks_or_null[k][i] : nullptr;
        }
        if (masks_or_null) return ec_masked_fused(o1, o2, o3, dt, pi, mi, scalars_or_null, n[i], out[i], out_mask_or_null[i], g->streams[i]);
        return ec_fused(o1, o2, o3, dt, pi, scalars_or_null, n[i], out[i], g->streams[i]);
    });
}

extern "C" ec_status ec_sharded_min_max(ec_shard_group* g, ec_dtype t, const void* const* p, const uint8_t* const* masks_or_null,
                                        const size_t* n, ec_value* mn, ec_value* mx) {
    ec_status st = check_group(g, "ec_sharded_min_max");
    if (st != EC_OK) return st;
    if (!p || !n || !mn || !mx) return set_error(EC_ERR_ARG, "ec_sharded_min_max: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    st = for_each_shard(g, [&](int i) {
        // an empty shard contributes the idempotent sentinels (T::MAX, T::MIN): src/buffer.rs:170
        ec_status s = ec_min_max_keys(t, p[i], masks_or_null ? masks_or_null[i] : nullptr, n[i], g->payload_dev[i], g->streams[i]);
        return s == EC_OK ? exchange(g, i, true) : s;
    });
    if (st != EC_OK) return st;
    int64_t keys[2] = {g->payload_host[0][0], g->payload_host[0][1]};
    if (g->comms.empty())  // host combine: element-wise MAX of {~key(min), key(max)}
        for (int i = 1; i < g->n; ++i) {
            if (g->payload_host[i][0] > keys[0]) keys[0] = g->payload_host[i][0];
            if (g->payload_host[i][1] > keys[1]) keys[1] = g->payload_host[i][1];
        }
    return ec_min_max_decode(t, keys, mn, mx);
}

extern "C" ec_status ec_sharded_counts(ec_shard_group* g, const uint8_t* const* masks, const size_t* n, uint64_t* n_true,
                                       uint64_t* n_false) {
    ec_status st = check_group(g, "ec_sharded_counts");
    if (st != EC_OK) return st;
    if (!masks || !n || !n_true || !n_false) return set_error(EC_ERR_ARG, "ec_sharded_counts: null argument");
    std::lock_guard<std::mutex> lk(g->call_mu);
    st = for_each_shard(g, [&](int i) {
        ec_status s = ec_mask_counts_device(masks[i], n[i], reinterpret_cast<uint64_t*>(g->payload_dev[i]), g->streams[i]);
        return s == EC_OK ? exchange(g, i, false) : s;
    });
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

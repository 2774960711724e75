// ec_collective.hip — the one exchange step of the sharded path, for hosts that hold an RCCL
// communicator themselves (a Rust or C++ host without torch.distributed): all-reduce of the
// 16-byte reduction payloads over xGMI.  librccl is resolved lazily (dlopen), so the library loads
// on machines without RCCL and single-GPU users never touch it.
//
//   min/max : keys2 = {~key(min), key(max)}  --ncclAllReduce(count 2, int64, MAX)-->  global keys
//   counts  : counts2 = {n_true, n_false}    --ncclAllReduce(count 2, uint64, SUM)--> global counts
//
// The payload is 16 B: latency-bound, link bandwidth irrelevant.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "ec_runtime.hpp"

using namespace ecd;

namespace {
using allreduce_fn = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
using errstr_fn = const char* (*)(ncclResult_t);
allreduce_fn g_allreduce = nullptr;
errstr_fn g_errstr = nullptr;
std::once_flag g_once;

void load_rccl() {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    g_allreduce = reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce"));
    g_errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
}

ec_status allreduce2(void* comm, void* buf, ncclDataType_t dt, ncclRedOp_t op, ec_stream stream, const char* what) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (!comm || !buf) return set_error(EC_ERR_ARG, "%s: null communicator or buffer", what);
    std::call_once(g_once, load_rccl);
    if (!g_allreduce) return set_error(EC_ERR_RCCL, "%s: librccl.so could not be loaded (%s)", what, dlerror() ? dlerror() : "symbol missing");
    const ncclResult_t r = g_allreduce(buf, buf, 2, dt, op, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return set_error(EC_ERR_RCCL, "%s: ncclAllReduce failed: %s", what, g_errstr ? g_errstr(r) : "?");
    return EC_OK;
}
}  // namespace

extern "C" ec_status ec_allreduce_min_max_keys(void* rccl_comm, int64_t* keys2_dev, ec_stream stream) {
    return allreduce2(rccl_comm, keys2_dev, ncclInt64, ncclMax, stream, "ec_allreduce_min_max_keys");
}

extern "C" ec_status ec_allreduce_counts(void* rccl_comm, uint64_t* counts2_dev, ec_stream stream) {
    return allreduce2(rccl_comm, counts2_dev, ncclUint64, ncclSum, stream, "ec_allreduce_counts");
}

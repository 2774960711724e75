// ec_collective.hip — the one exchange step of the sharded path: all-reduce of the 16-byte reduction
// payloads over xGMI, plus the communicator bootstrap a host without torch.distributed needs
// (a Rust or C++ program; ec_sharded.hip's in-process shard group).  librccl is resolved lazily
// (dlopen), so the library loads on machines without RCCL and single-GPU users never touch it.
//
//   min/max : keys2 = {~key(min), key(max)}  --ncclAllReduce(count 2, int64, MAX)-->  global keys
//   counts  : counts2 = {n_true, n_false}    --ncclAllReduce(count 2, uint64, SUM)--> global counts
//
// The payload is 16 B: latency-bound, link bandwidth irrelevant.
//
// Communicators come in the two shapes RCCL offers:
//   one process per GPU   rank 0: ec_comm_get_unique_id -> the host program hands the 128 bytes to the other
//                         ranks (file, socket, MPI, env) -> every rank: ec_comm_init_rank on its own device;
//   one process, n GPUs   ec_comm_init_all(devices, n, comms).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "ec_collective.hpp"
#include "ec_runtime.hpp"

using namespace ecd;

static_assert(sizeof(ncclUniqueId) == sizeof(ec_comm_uid), "ec_comm_uid mirrors ncclUniqueId (128 bytes)");

namespace ecd {

static Rccl g_rccl;
static std::once_flag g_once;
static std::string g_load_error;

static void load_rccl() {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char* e = dlerror();
        g_load_error = e ? e : "dlopen failed";
        return;
    }
#define EC_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name))
    EC_SYM(all_reduce, "ncclAllReduce");
    EC_SYM(error_string, "ncclGetErrorString");
    EC_SYM(get_unique_id, "ncclGetUniqueId");
    EC_SYM(comm_init_rank, "ncclCommInitRank");
    EC_SYM(comm_init_all, "ncclCommInitAll");
    EC_SYM(comm_destroy, "ncclCommDestroy");
    EC_SYM(comm_abort, "ncclCommAbort");
    EC_SYM(group_start, "ncclGroupStart");
    EC_SYM(group_end, "ncclGroupEnd");
#undef EC_SYM
    if (!g_rccl.all_reduce || !g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_init_all ||
        !g_rccl.comm_destroy || !g_rccl.group_start || !g_rccl.group_end) {
        g_load_error = "librccl.so lacks one of the ncclAllReduce / ncclCommInit* / ncclGroup* symbols";
        g_rccl = Rccl{};
    }
}

const Rccl* rccl(const char* what) {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.all_reduce) {
        set_error(EC_ERR_RCCL, "%s: librccl.so could not be loaded (%s)", what, g_load_error.c_str());
        return nullptr;
    }
    return &g_rccl;
}

ec_status check_rccl(int r, const char* what) {
    if (r == ncclSuccess) return EC_OK;
    return set_error(EC_ERR_RCCL, "%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(static_cast<ncclResult_t>(r)) : "?");
}

}  // namespace ecd

namespace {
ec_status allreduce2(void* comm, void* buf, ncclDataType_t dt, ncclRedOp_t op, ec_stream stream, const char* what) {
    ec_status st = ensure_ready();
    if (st != EC_OK) return st;
    if (!comm || !buf) return set_error(EC_ERR_ARG, "%s: null communicator or buffer", what);
    const Rccl* R = rccl(what);
    if (!R) return EC_ERR_RCCL;
    return check_rccl(R->all_reduce(buf, buf, 2, dt, op, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)), what);
}
}  // namespace

extern "C" ec_status ec_allreduce_min_max_keys(ec_comm comm, int64_t* keys2_dev, ec_stream stream) {
    return allreduce2(comm, keys2_dev, ncclInt64, ncclMax, stream, "ec_allreduce_min_max_keys");
}

extern "C" ec_status ec_allreduce_counts(ec_comm comm, uint64_t* counts2_dev, ec_stream stream) {
    return allreduce2(comm, counts2_dev, ncclUint64, ncclSum, stream, "ec_allreduce_counts");
}

// ------------------------------------------------------------------ communicator bootstrap
extern "C" ec_status ec_comm_get_unique_id(ec_comm_uid* uid) {
    if (!uid) return set_error(EC_ERR_ARG, "ec_comm_get_unique_id: null out");
    const Rccl* R = rccl("ec_comm_get_unique_id");
    if (!R) return EC_ERR_RCCL;
    ncclUniqueId id;
    ec_status st = check_rccl(R->get_unique_id(&id), "ncclGetUniqueId");
    if (st != EC_OK) return st;
    std::memcpy(uid, &id, sizeof id);
    return EC_OK;
}

extern "C" ec_status ec_comm_init_rank(const ec_comm_uid* uid, int32_t n_ranks, int32_t rank, ec_comm* comm) {
    if (!uid || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return set_error(EC_ERR_ARG, "ec_comm_init_rank: rank %d of %d", int(rank), int(n_ranks));
    ec_status st = ensure_ready();  // the communicator is bound to the calling thread's current library device
    if (st != EC_OK) return st;
    const Rccl* R = rccl("ec_comm_init_rank");
    if (!R) return EC_ERR_RCCL;
    ncclUniqueId id;
    std::memcpy(&id, uid, sizeof id);
    ncclComm_t c = nullptr;
    st = check_rccl(R->comm_init_rank(&c, n_ranks, id, rank), "ncclCommInitRank");
    if (st != EC_OK) return st;
    *comm = c;
    return EC_OK;
}

extern "C" ec_status ec_comm_init_all(const int32_t* devices, int32_t n, ec_comm* comms) {
    if (!devices || !comms || n < 1) return set_error(EC_ERR_ARG, "ec_comm_init_all: null argument or n < 1");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j])
                return set_error(EC_ERR_ARG, "ec_comm_init_all: device %d listed twice (RCCL needs one rank per GPU)", int(devices[i]));
    int32_t before = -1;
    const bool had = ec_get_device(&before) == EC_OK;
    for (int i = 0; i < n; ++i) {  // every device of the clique gets its runtime state
        ec_status st = ec_init(devices[i]);
        if (st != EC_OK) return st;
    }
    if (had) (void)ec_set_device(before);
    const Rccl* R = rccl("ec_comm_init_all");
    if (!R) return EC_ERR_RCCL;
    std::vector<ncclComm_t> cs(n, nullptr);
    std::vector<int> devs(devices, devices + n);
    ec_status st = check_rccl(R->comm_init_all(cs.data(), n, devs.data()), "ncclCommInitAll");
    if (st != EC_OK) return st;
    for (int i = 0; i < n; ++i) comms[i] = cs[i];
    return EC_OK;
}

extern "C" ec_status ec_comm_destroy(ec_comm comm) {
    if (!comm) return EC_OK;
    const Rccl* R = rccl("ec_comm_destroy");
    if (!R) return EC_ERR_RCCL;
    return check_rccl(R->comm_destroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

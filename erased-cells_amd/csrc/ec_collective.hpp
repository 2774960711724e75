// ec_collective.hpp — the lazily resolved RCCL entry points shared by ec_collective.hip and ec_sharded.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "erased_cells.h"

namespace ecd {

struct Rccl {
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_init_all)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_abort)(ncclComm_t) = nullptr;  // optional: used only to unblock a poisoned shard group
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
};

// nullptr (and the thread's error text set, EC_ERR_RCCL) when librccl cannot be loaded.
const Rccl* rccl(const char* what);
ec_status check_rccl(int nccl_result, const char* what);

}  // namespace ecd

// ec_hostpipe.hpp — page-locking of caller-owned host ranges for the host-to-host pipelines (ec_hostpipe.hip, ec_sharded.hip).
#pragma once

#include <cstddef>
#include <cstdint>

namespace ecd {

// Page-locks a host range for the life of the object unless the caller already did.  Calls that run at the same time may
// share operands (the same numpy array in two threads; the row-blocks of one raster on the launch threads of a shard
// group): the registrations the library makes are counted in one table, so that the call that finishes first does not
// unregister pages another call is still copying from, and a range inside a registration in flight shares it.  A range that
// only partly overlaps one waits for it (hipHostRegister refuses overlapping ranges).
struct Pinned {
    uintptr_t base = 0;  // key of the table entry this object holds a reference on (0: none)
    void pin(const void* ptr, size_t bytes);
    ~Pinned();
    Pinned() = default;
    Pinned(const Pinned&) = delete;
    Pinned& operator=(const Pinned&) = delete;
};

}  // namespace ecd

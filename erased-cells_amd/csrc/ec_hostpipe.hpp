// ec_hostpipe.hpp — page-locking of caller-owned host ranges for the host-to-host pipelines (ec_hostpipe.hip, ec_sharded.hip).
#pragma once

#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace ecd {

// Page-locks the host ranges of ONE call for the life of the object, unless the caller already did.
//
// hipMemcpyAsync refuses a range that lies only partly inside a registration, and hipHostRegister refuses a range that overlaps
// one — so the ranges of a call are first widened to whole pages and merged (two windows of one array as two operands, two
// small arrays on one page), and every registration the library makes is counted in one table: calls that run at the same
// time may share operands (the same array in two threads; the row-blocks of a raster on the launch threads of a shard group,
// inside the registration their caller made for the whole raster), and the call that finishes first does not unregister
// pages another is still copying from.  A range that partly overlaps a registration of ANOTHER call in flight waits for
// that call — holding nothing while it waits, all of a call's ranges are taken at once.
//
// A call that copies WITHOUT page-locking (the small form of ec_host_expr: the runtime's pageable path) enters its ranges too
// (`use_all`): the HIP runtime looks a host pointer's registration up when a copy is queued, and a registration that another
// thread makes or removes under a copy in flight took the process down ("pure virtual method called") in a soak of the round's
// entry points from five host threads.  Ranges in use without a registration make an overlapping registration wait; a range
// inside a registration in flight shares it.
class PinSet {
    struct Held { uintptr_t base; size_t bytes; bool registered; bool refused = false; };
    std::vector<Held> held_;  // table entries this object holds a reference on

public:
    void pin_all(const std::vector<std::pair<const void*, size_t>>& ranges);
    void use_all(const std::vector<std::pair<const void*, size_t>>& ranges);
    ~PinSet();
    PinSet() = default;
    PinSet(const PinSet&) = delete;
    PinSet& operator=(const PinSet&) = delete;
};

}  // namespace ecd

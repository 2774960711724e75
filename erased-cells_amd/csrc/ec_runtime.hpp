// ec_runtime.hpp — process-wide state shared by the translation units of
// liberased_cells_hip.so: error reporting, launch-shape knobs, reduction scratch.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>

#include "erased_cells.h"

namespace ecd {

// Compile-time launch shape of the binary-arithmetic kernels; chosen from the
// A/B runs recorded in profiles/ (tools/tune_binop.hip).
constexpr int kBinopU = 2;         // chunks of 128 cells per wave per tile (U = 2 beats 8 by ≈6 %: tune_binop_v4/v5.log)
#ifndef EC_NT_STORE
#define EC_NT_STORE 1  // build-time A/B switch (make EXTRA=-DEC_NT_STORE=0)
#endif
#ifndef EC_NT_LOAD
#define EC_NT_LOAD 1
#endif
constexpr bool kNtStore = EC_NT_STORE;    // streaming f64 output: 2.1 GB ≫ 256 MiB Infinity Cache
constexpr bool kNtLoad = EC_NT_LOAD;     // +6 % on the divide (tune_binop_v2.log: 6270 -> 6666 GB/s)
constexpr int kReduceU = 8;   // 16-byte loads in flight per lane of a reduction tile (ec_reduce_kernels.hpp)
constexpr int kMaxReduceBlocks = 4096;

struct Tuning;
Tuning& tuning();
// Process-wide knobs (ec_tune_set).  Each field is an atomic word: a knob may be turned while other host threads
// launch — a launch sees the old or the new value of each knob, never a torn one; both are valid launch shapes.
struct Tuning {
    std::atomic<int> binop_variant{-1};  // -1 = by rule (ec_binop_tu.hpp: LDS-staged for an 8-byte operand against one of <= 4 bytes), 0 = always direct narrow loads, 1 = LDS-staged wherever an operand can be staged
    std::atomic<int> reduce_bpc{0};     // workgroups per CU for reductions; 0 = each launch shape's own default (4 x 512 threads)
    std::atomic<int> reduce_shape{0};   // min_max launch shape A/B (ec_abi.hip launch_min_max): 0 = 512 thr x 8 loads (default)
    std::atomic<int> map_u{2};          // 16-B groups per lane per tile for the map kernels (1, 2 or 4)
    std::atomic<int> peel{1};              // leading-cell peel of the binop/fused kernels: 0 off, 1 for 1-byte operands, 2 also for 2-byte ones
    std::atomic<int> unaligned_vector{1};  // 1 = vector kernels at any cell offset (gfx950 unaligned global access);
                                           // 0 = pointers that are not 16-byte aligned run the cell-wise kernels
    std::atomic<int> fused_mixed{1};       // 1 = one-pass typed-load kernels for fused chains over mixed cell types; 0 = convert, then fuse
    std::atomic<int64_t> mall_mb{256};     // Infinity Cache budget of cache_plan(): operand streams that fit it together are loaded with
                                           // the default cache policy, all others non-temporal; 0 = every stream non-temporal
    std::atomic<int> inject_shard_failure{0};  // test hook: shard index + 1 whose NEXT fire-and-forget job of a shard group reports
                                               // EC_ERR_HIP instead of launching (exercises the deferred-error path); 0 = off
    std::atomic<int> inject_pin_refusal{0};    // test hook: 1 = PinSet::pin_all treats every hipHostRegister as refused, so the
                                               // pageable fallback of the host-to-host pipelines is exercised
    std::atomic<int> expr_jit{1};  // expression programs compiled at run time (ec_expr_jit.hpp): 0 never, 1 in the background once a
                                   // program has interpreted 2^31 cell-steps, 2 on the calling thread at first sight
    std::atomic<int> expr_fixed{1};  // 1 (default): a program of the ahead-of-time catalogue (ec_expr_fixed.hpp) runs as its built-in straight-line
                                     // kernel; 0: never (the interpreter / the compiled form serve it — the comparison path of the tests)
    std::atomic<int> write_lds_kb{64};  // LDS reserved per workgroup of a PURE-WRITE launch (ec_fill): caps the workgroups resident
                                        // per CU (64 KiB: two of 160 KiB).  A write-only stream runs faster from few resident waves — 0.85 of the HBM
                                        // peak at full occupancy, 0.88-0.90 at 2-3 workgroups per CU, 0.93 with write-through stores on top
                                        // (profiles/r04/tune_store_v2.log); any launch that also loads needs its occupancy and gets none.  0 = no cap
    // Occupancy caps (unused LDS reserved per workgroup) for kernels that load with 16 bytes per lane: such loads keep enough bytes in flight
    // from fewer waves, and the store stream likes fewer (profiles/r04/write_heavy_caps.md, rotating sets, 16384²): f64 ∘ f64 0.778 -> 0.793 at
    // 48 KiB (3 workgroups per CU), f64 ∘ scalar 0.806 -> 0.828 at 32 KiB (5).  Every kernel with a NARROW operand loses under any cap
    // (u8 ∘ scalar 0.75 -> 0.52 at 32 KiB) and takes none.  -1 = that rule; >= 0 forces the reservation for every launch of the family.
    std::atomic<int> binop_lds_kb{-1};
    std::atomic<int> scalar_lds_kb{-1};
    std::atomic<int> map_lds_kb{0};     // experiment knob for the map launches that load (convert, neg, mask_select …): no rule adopted
    std::atomic<int> fused_lds_kb{0};   // the same reservation for the one-pass kernels (k_fused_any, k_expr, k_expr_fixed): an experiment knob (profiles/r04/fused_caps.md)
    std::atomic<int> counts_one_launch{1};  // Mask::counts in ONE launch: every workgroup adds (1 << 40 | its count) to one 64-bit word of the stream's
                                            // scratch with a returning atomic, the workgroup that reads grid - 1 in the upper bits owns the total,
                                            // writes the result and zeroes the word — one device-scope round trip behind the last load instead of a
                                            // second launch (0: partials + finalize kernel, the form of rounds 1-3; 1: below 2^29 cells; 2: always)
    std::atomic<int> cache_force{-1};  // A/B hook: >= 0 replaces cache_plan()'s answer by these bits for every launch (profiles/r04/cache_plan_ab.md)
    std::atomic<int64_t> pool_keep_mb{32768};  // release threshold of the library's stream-ordered pool (per device)
};

int device_cus();      // CU count of the device bound by the last ensure_ready() on this thread
int current_device();  // that device's index

ec_status ensure_ready();  // EC_ERR_NOT_INITIALIZED unless ec_init ran; binds the calling thread to its library device
ec_status set_error(ec_status code, const char* fmt, ...);
ec_status set_error_text(ec_status code, const std::string& text);
const std::string& last_error_text();
ec_status set_narrowing(int src, int dst);
std::atomic<int64_t>& lds_rule_launches();  // binop launches that took the LDS-staged variant by rule (ec_stat_get "binop_lds_rule_launches")
ec_status check_launch(const char* what);
ec_status check_hip(hipError_t e, const char* what);

// May the vector kernels run on these pointers?  Their loads and stores are declared under-aligned
// (ec_device.hpp nt_load/nt_store), so the answer is yes at any cell offset unless the
// "unaligned_vector" knob is turned off.
inline bool aligned16(const void* a, const void* b, const void* c) {
    return tuning().unaligned_vector ||
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15u) == 0;
}
inline bool aligned_to(const void* a, size_t bytes) {
    return tuning().unaligned_vector || reinterpret_cast<uintptr_t>(a) % bytes == 0;
}

// Leading cells (0 or 1) the direct binop / fused kernels peel so that the 2-cell loads of 1-byte operands start
// on even addresses: a u8x2 load at an odd address costs ≈6 % of the kernel, more than the 8-byte shift of the
// f64 pair stores the peel causes (u8∘u16 at an odd offset: 0.75 of peak unpeeled, 0.78–0.80 peeled).  Pair
// loads of 2-byte cells at 2 mod 4 cost nothing, so they are left alone — peeling for them only misaligns the
// output (u16∘u16: 0.78 unpeeled, 0.70–0.73 peeled; profiles/r01/peel_ab.md).
inline unsigned peel_cost(const void* p, size_t size, unsigned h) {
    const bool counts = size == 1 || (size == 2 && tuning().peel >= 2);
    return counts ? static_cast<unsigned>(size) * static_cast<unsigned>(((reinterpret_cast<uintptr_t>(p) / size) + h) & 1) : 0u;
}
inline unsigned peel_head(const void* l, size_t lsize, const void* r, size_t rsize, size_t n) {
    if (n < 2 || !tuning().peel) return 0;
    const unsigned c0 = peel_cost(l, lsize, 0) + (r ? peel_cost(r, rsize, 0) : 0);
    const unsigned c1 = peel_cost(l, lsize, 1) + (r ? peel_cost(r, rsize, 1) : 0);
    return c1 < c0 ? 1u : 0u;
}

// Load policy of a launch (policy_arms(), ec_device.hpp): bit k set = operand stream k is loaded with the default cache
// policy because it may still be — or is worth leaving — in the 256 MiB Infinity Cache; clear = non-temporal.  Streams
// are admitted smallest first while they fit the budget TOGETHER (two 256 MiB operands would only evict each other); a
// stream bigger than the cache is always streamed.  The figures behind the rule (MI355X, 16384² cells, profiles/r03/
// tune_nt_width_rotating.log): the u8 ÷ u16 divide with its 256 MiB u8 operand cacheable runs at 0.856 of the HBM peak
// when that operand was touched by the previous launch and 0.792 when it was not; with every load nt, 0.80 either way.
// Big streams want nt: a 2 GiB read-only stream reaches 0.865 nt against 0.76 cacheable.
//
// `value_out_bytes` (round 4): the f64 result the launch streams out, for the kernels that write one (binop, fused, expr).  With it
// a second rule applies: if a stream is LEFT OUT that is no bigger than one that was admitted — two equal operands of which only one
// fits — nothing is admitted.  Measured (profiles/r04/cache_plan_ab.md): u8 ∘ u8 → f64 at 16384² with one of its two 256 MiB operands
// cacheable runs 1.5 % (add) to 7 % (divide) SLOWER in the one-set loop than with every load nt — at any relative placement of the
// two buffers (tools/cache_conflict_probe.py), with either one admitted — while u8 ∘ u16 with its u8 operand cacheable runs 5-8 %
// faster, and the 1 B/cell kernels (mask_and with one of two masks cacheable: 0.93 against 0.80) keep their gain: they pass 0 here.
// An operand that is never read again costs at most 0.6 % under either rule (the rotating-set column of the same table).
inline unsigned cache_plan(const size_t* bytes, int n, size_t value_out_bytes = 0) {
    const int forced = tuning().cache_force.load();
    if (forced >= 0) return static_cast<unsigned>(forced) & ((1u << (n < 8 ? n : 8)) - 1u);
    const size_t budget = static_cast<size_t>(tuning().mall_mb.load()) << 20;
    unsigned plan = 0;
    size_t used = 0, largest_taken = 0;
    bool taken[8] = {false, false, false, false, false, false, false, false};
    for (int round = 0; round < n && n <= 8; ++round) {
        int best = -1;
        for (int k = 0; k < n; ++k)
            if (!taken[k] && bytes[k] > 0 && (best < 0 || bytes[k] < bytes[best])) best = k;
        if (best < 0 || used + bytes[best] > budget) break;
        taken[best] = true;
        used += bytes[best];
        if (bytes[best] > largest_taken) largest_taken = bytes[best];
        plan |= 1u << best;
    }
    if (value_out_bytes > 0 && plan != 0)
        for (int k = 0; k < n && n <= 8; ++k)
            if (!taken[k] && bytes[k] > 0 && bytes[k] <= largest_taken) return 0;  // an equal (or smaller) peer does not fit: admit none
    return plan;
}

// Leading cells a reduction peels so that its 16-byte loads start 16-byte aligned (0 when the window is shorter).
inline unsigned reduce_head(const void* p, size_t cell_size, size_t n) {
    if (!tuning().unaligned_vector) return 0;
    const size_t h = ((16 - reinterpret_cast<uintptr_t>(p) % 16) % 16) / cell_size;
    return h <= n ? static_cast<unsigned>(h) : 0u;
}

// Element-wise kernels: one workgroup per tile.
inline unsigned grid_for(size_t tiles) {
    if (tiles < 1) tiles = 1;
    return static_cast<unsigned>(tiles < size_t(0x7fffffff) ? tiles : size_t(0x7fffffff));
}
// Grid-stride kernels (cell-wise fallbacks, reductions): at most bpc workgroups per CU.
inline unsigned grid_capped(size_t blocks, int bpc) {
    if (blocks < 1) blocks = 1;
    const size_t cap = size_t(device_cus()) * size_t(bpc > 0 ? bpc : 8);
    return static_cast<unsigned>(blocks < cap ? blocks : cap);
}

// Per-(device, stream) scratch for reduction partials (device) and results (pinned host).  Launches that use the
// partials are ordered by the stream itself; `mu` serialises the synchronous-result entry points (ec_min_max,
// ec_mask_counts, ec_first_difference), which read `host` after waiting for the stream — two host threads sharing
// one stream therefore take turns instead of racing on the four pinned words.
struct ScratchOwner;  // frees the three allocations when the last Scratch copy that names them is gone (ec_runtime.hip)
struct Scratch {
    std::shared_ptr<ScratchOwner> owner;  // every copy handed out by get_scratch() shares ownership: releasing or recycling a
                                          // stream's entry only drops the TABLE's reference, so a host thread that is
                                          // inside ec_min_max / ec_mask_counts / ec_first_difference with this scratch
                                          // (its `mu` locked, its kernels queued) keeps valid memory until it returns
    int64_t* dev = nullptr;    // 2*kMaxReduceBlocks partials + 4 result words + 4 accumulator words (zero between kernels)
    int64_t* host = nullptr;   // 4 words, pinned (coherent): the synchronous-result entry points let the last kernel write
                               // its result straight into them — no device-to-host copy is queued behind the kernel
    int64_t* host_dev = nullptr;  // the same words as the device addresses them
    std::mutex* mu = nullptr;
    int64_t* dev_result() const { return dev + 2 * kMaxReduceBlocks; }
    int64_t* dev_acc() const { return dev + 2 * kMaxReduceBlocks + 4; }
};
ec_status get_scratch(hipStream_t s, Scratch* out);

// binary arithmetic, one translation unit per op
template <int OP>
ec_status dispatch_binop(int lt, const void* l, int rt, const void* r, size_t n, double* out, hipStream_t s);
template <int OP>
ec_status dispatch_masked_binop(int lt, const void* l, const uint8_t* lm, int rt, const void* r, const uint8_t* rm,
                                size_t n, double* out, uint8_t* om, hipStream_t s);
template <int OP>
ec_status dispatch_binop_scalar(int lt, const void* l, double rhs, size_t n, double* out, hipStream_t s);

}  // namespace ecd

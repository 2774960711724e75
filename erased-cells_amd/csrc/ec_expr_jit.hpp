// ec_expr_jit.hpp — expression programs compiled at run time (hiprtc): the value phase of ec_expr as straight-line code.
//
// k_expr (ec_expr.hpp) interprets a program per tile and is bound by instruction issue: ≈ 45 scalar instructions of
// decode per step and wave (profiles/r03/expr_kernel.md).  A program is launch-uniform, so the decode can be done ONCE, on
// the host: ec_expr_jit.hip writes the HIP source of a kernel that IS the program — typed loads of the streams' own cell
// types, the steps as straight-line f64 operations in the program's order (so the cells are the interpreter's and the
// eager chain's, bit for bit), the same NaN rule, tile shape, load policy and nt stores — compiles it with hiprtc for the
// device's architecture and caches the module per (program, cell types, load policy).
//
// The interpreter stays the reference path and the fallback: a program's first launches run on it while a background
// thread compiles; libhiprtc is resolved lazily (dlopen) and a machine without it simply keeps interpreting (counted:
// ec_stat_get("expr_jit_failures")).  ec_tune_set("expr_jit", …): 0 = never compile, 1 = compile in the background once a
// program has interpreted 2^31 cell-steps (default), 2 = compile on the calling thread at first sight (tests, bench).
#pragma once

#include <string>

#include "ec_expr.hpp"

namespace ecd {

// HIP source of the kernel `ec_expr_jit` for the program in `ea` (prog / dt / nstreams / nsteps / nmask / cacheable are read).
std::string expr_jit_source(const ExprArgs& ea, bool reduce = false);

// Compile `source` for `arch` ("gfx950") with hiprtc; the code object in `code`, the compiler's log in `log`.
ec_status expr_jit_compile(const std::string& source, const std::string& arch, std::string* code, std::string* log);

// Run the program (values and, for a masked call, the AND of the masks) through its compiled kernel if one is (or, in mode 2,
// can be made) ready.
// *launched = false and EC_OK when the caller should interpret (not compiled yet, compiling, hiprtc missing, capture in
// progress before the module was loaded).
// `keys2` != nullptr: the REDUCE variant — nothing is stored; the kernel folds {~key(min), key(max)} of the valid cells' values
// (order keys of f64 under total_cmp, ec_device.hpp order_key) into keys2[0..1] with atomic max (the caller initialises them).
ec_status expr_jit_launch(const ExprArgs& ea, size_t n, double* out, uint8_t* out_mask, int64_t* keys2, hipStream_t s, bool* launched);

int64_t expr_jit_stat(const char* key, bool* known);
void expr_jit_release();  // ec_shutdown: unload every loaded module

}  // namespace ecd

"""Typed lockstep of the Rust `extern "C"` block with the C header.

The Rust crate cannot be compiled in this image (no rustc), so a wrong parameter type in
`erased-cells_amd/rust/erased-cells-hip/src/ffi.rs` — `*const u8` where the header says `const void *const *` — would
only show up as a crash on the first machine that builds it.  This test parses every prototype of
`include/erased_cells.h` and every `pub fn` of the extern block and compares them TYPE BY TYPE, position by position,
through a fixed C -> Rust map (pointer depth and the constness of every level included), plus the return types, the
typedefs, the constants the crate mirrors, the callback type `ec_shard_fn` and the layout of `ec_value` (against what gcc
says).  A self-check mutates declarations in memory and expects the comparison to notice.  No GPU, no reference tree needed.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "erased_cells.h")
FFI_RS = os.path.join(ROOT, "erased-cells_amd", "rust", "erased-cells-hip", "src", "ffi.rs")

C_BASE = {  # C base type -> Rust spelling
    "void": "c_void", "char": "c_char", "double": "f64", "float": "f32", "size_t": "usize", "int": "i32",
    "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64",
    "int8_t": "i8", "int16_t": "i16", "int32_t": "i32", "int64_t": "i64",
    # the ABI's own names exist on both sides (their definitions are compared separately)
    "ec_status": "ec_status", "ec_dtype": "ec_dtype", "ec_op": "ec_op", "ec_stream": "ec_stream", "ec_comm": "ec_comm",
    "ec_value": "ec_value", "ec_comm_uid": "ec_comm_uid", "ec_shard_group": "ec_shard_group", "ec_shard_fn": "ec_shard_fn",
    "ec_expr_step": "ec_expr_step",
}


def c_type_to_rust(decl: str, is_param: bool = True) -> str:
    """`const void *const p[4]` -> `*const *const c_void`.  `decl` is one parameter (name optional) or a return type."""
    decl = decl.strip()
    array = False
    m = re.search(r"\[\s*\d*\s*\]\s*$", decl)
    if m:  # a parameter `T x[N]` is `T *x`
        array, decl = True, decl[:m.start()].strip()
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", decl)
    if "struct" in toks:
        toks.remove("struct")
    # the parameter's name: a trailing identifier that is neither a qualifier nor the (only) base type
    names = [t for t in toks if t not in ("const", "*")]
    if is_param and len(names) > 1 and toks[-1] not in ("const", "*"):
        toks = toks[:-1]
    # base (with its const), then one (pointer, const-of-that-pointer) per '*'
    i, base, base_const = 0, None, False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        else:
            assert base is None, f"two base types in {decl!r}"
            base = toks[i]
        i += 1
    assert base in C_BASE, f"unmapped C type {base!r} in {decl!r}"
    levels = []  # constness of each pointer OBJECT, innermost first
    while i < len(toks):
        assert toks[i] == "*", decl
        i += 1
        own_const = False
        while i < len(toks) and toks[i] == "const":
            own_const = True
            i += 1
        levels.append(own_const)
    rust, pointee_const = C_BASE[base], base_const
    for own_const in levels:
        rust = ("*const " if pointee_const else "*mut ") + rust
        pointee_const = own_const
    if array:
        rust = ("*const " if pointee_const else "*mut ") + rust
    return rust


def _split_params(s: str):
    s = s.strip()
    return [] if s in ("", "void") else [p.strip() for p in s.split(",")]


def header_prototypes():
    """{name: (return type as Rust, [param types as Rust])} and the typedef map of the header."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    body = re.sub(r"\{[^{}]*\}", "", re.sub(r"\{[^{}]*\}", "", text))  # drop enum / struct / union bodies (one nesting level)
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^(?:extern \"C\" )?((?:const )?[A-Za-z_][A-Za-z0-9_]*[ *]+)(ec_[a-z0-9_]+) ?\(([^()]*)\)$", stmt)
        if not m or stmt.startswith("typedef"):
            continue
        ret, name, params = m.group(1), m.group(2), m.group(3)
        protos[name] = (c_type_to_rust(ret, is_param=False), [c_type_to_rust(p) for p in _split_params(params)])
    return protos, text


def rust_prototypes(src=None):
    src = open(FFI_RS).read() if src is None else src
    src = re.sub(r"//[^\n]*", "", src)
    block = src[src.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (ec_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        ptypes = []
        for p in _split_params(params):
            assert ":" in p, f"{name}: parameter without a type: {p!r}"
            ptypes.append(re.sub(r"\s+", " ", p.split(":", 1)[1].strip()))
        out[name] = (re.sub(r"\s+", " ", ret.strip()) if ret else "()", ptypes)
    return out, src


def mismatches(c_protos, r_protos):
    bad = []
    for name in sorted(set(c_protos) | set(r_protos)):
        if name not in c_protos or name not in r_protos:
            bad.append(f"{name}: declared only in {'the header' if name in c_protos else 'ffi.rs'}")
            continue
        (cr, cp), (rr, rp) = c_protos[name], r_protos[name]
        if cr != rr:
            bad.append(f"{name}: returns {cr} in the header, {rr} in ffi.rs")
        if len(cp) != len(rp):
            bad.append(f"{name}: {len(cp)} parameters in the header, {len(rp)} in ffi.rs")
            continue
        for i, (a, b) in enumerate(zip(cp, rp)):
            if a != b:
                bad.append(f"{name}: parameter {i} is {a} in the header, {b} in ffi.rs")
    return bad


def test_c_declarator_translation():
    t = c_type_to_rust
    assert t("ec_op op") == "ec_op"
    assert t("const void *l") == "*const c_void"
    assert t("double *out") == "*mut f64"
    assert t("void **dptr") == "*mut *mut c_void"
    assert t("const void *const *l") == "*const *const c_void"
    assert t("double *const *out") == "*const *mut f64"
    assert t("void *const *dptrs") == "*const *mut c_void"
    assert t("const ec_dtype dt[4]") == "*const ec_dtype"
    assert t("const void *const p[4]") == "*const *const c_void"
    assert t("const void *const *const p[4]") == "*const *const *const c_void"
    assert t("const uint8_t *const *const masks_or_null[4]") == "*const *const *const u8"
    assert t("const int64_t keys2_host[2]") == "*const i64"
    assert t("ec_shard_group **out") == "*mut *mut ec_shard_group"
    assert t("const ec_shard_group *g") == "*const ec_shard_group"
    assert t("const char *", is_param=False) == "*const c_char"
    assert t("size_t", is_param=False) == "usize"
    assert t("void", is_param=False) == "c_void"


def test_every_parameter_type_of_the_extern_block_matches_the_header():
    c_protos, _ = header_prototypes()
    r_protos, _ = rust_prototypes()
    assert len(c_protos) >= 80, f"only {len(c_protos)} prototypes parsed from the header"
    bad = mismatches(c_protos, r_protos)
    assert not bad, "ffi.rs and include/erased_cells.h disagree:\n  " + "\n  ".join(bad)


def test_the_comparison_notices_a_wrong_type():
    """Alter single declarations of ffi.rs in memory: each alteration must be reported."""
    c_protos, _ = header_prototypes()
    src = open(FFI_RS).read()
    for old, new, expect in (
        ("p: *const *const c_void,\n                    scalars_or_null", "p: *const u8,\n                    scalars_or_null", "ec_fused: parameter 4"),
        ("pub fn ec_binop(op: ec_op, lt: ec_dtype, l: *const c_void", "pub fn ec_binop(op: ec_op, lt: ec_dtype, l: *mut c_void", "ec_binop: parameter 2"),
        ("n: *const usize, out: *const *mut f64) -> ec_status;", "n: *const usize, out: *mut *mut f64) -> ec_status;", "ec_sharded_binop: parameter 7"),
        ("pub fn ec_size_of(t: ec_dtype) -> usize;", "pub fn ec_size_of(t: ec_dtype) -> u32;", "ec_size_of: returns"),
        ("pub fn ec_pool_trim(keep_bytes: usize)", "pub fn ec_pool_trim(keep_bytes: u32)", "ec_pool_trim: parameter 0"),
    ):
        assert old in src, f"the self-check's anchor is gone from ffi.rs: {old!r}"
        r_protos, _ = rust_prototypes(src.replace(old, new, 1))
        bad = mismatches(c_protos, r_protos)
        assert any(b.startswith(expect) for b in bad), (expect, bad)


def test_typedefs_constants_and_the_callback_type_match():
    _, hdr = header_prototypes()
    _, rs = rust_prototypes()
    c_typedefs = {m.group(2): c_type_to_rust(m.group(1), is_param=False)
                  for m in re.finditer(r"typedef\s+([A-Za-z_0-9 ]+?\s*\**)\s*(ec_[a-z_]+)\s*;", hdr)}
    r_typedefs = {m.group(1): re.sub(r"\s+", " ", m.group(2).strip()) for m in re.finditer(r"pub type (ec_[a-z_]+)\s*=\s*([^;]+);", rs)
                  if "fn(" not in m.group(2)}
    for name in ("ec_status", "ec_dtype", "ec_op", "ec_stream", "ec_comm"):
        assert c_typedefs[name] == r_typedefs[name], (name, c_typedefs[name], r_typedefs[name])
    # the callback: typedef ec_status (*ec_shard_fn)(int32_t shard, int32_t device, ec_stream stream, void *user);
    m = re.search(r"typedef\s+(\w+)\s*\(\s*\*\s*ec_shard_fn\s*\)\s*\(([^)]*)\)\s*;", hdr)
    c_cb = (c_type_to_rust(m.group(1), is_param=False), [c_type_to_rust(p) for p in _split_params(m.group(2))])
    m = re.search(r'pub type ec_shard_fn\s*=\s*extern "C" fn\(([^)]*)\)\s*->\s*([^;]+);', rs)
    r_cb = (m.group(2).strip(), [p.split(":", 1)[1].strip() for p in _split_params(m.group(1))])
    assert c_cb == r_cb, (c_cb, r_cb)
    # enum constants the crate mirrors
    c_consts = {k: int(v) for k, v in re.findall(r"\b(EC_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr)}
    r_consts = {k: int(v) for k, v in re.findall(r"pub const (EC_[A-Z0-9_]+)\s*:\s*\w+\s*=\s*(-?\d+)\s*;", rs)}
    assert r_consts, "no constants parsed from ffi.rs"
    for k, v in r_consts.items():
        assert c_consts.get(k) == v, f"{k}: {v} in ffi.rs, {c_consts.get(k)} in the header"
    for k in ("EC_OK", "EC_ERR_NARROWING", "EC_ADD", "EC_SUB", "EC_MUL", "EC_DIV", "EC_GROUP_HOST_COMBINE", "EC_GROUP_BLOCKING_ISSUE"):
        assert k in r_consts, f"ffi.rs lacks {k}"


def test_ec_value_layout_is_what_the_c_compiler_says(tmp_path):
    """`#[repr(C)] struct ec_value { dtype: u8, pad_: [u8; 7], bits: u64 }` against sizeof / offsetof from gcc, and the
    128-byte communicator id."""
    rs = open(FFI_RS).read()
    m = re.search(r"#\[repr\(C\)\]\s*#\[derive\([^)]*\)\]\s*pub struct ec_value\s*\{([^}]*)\}", rs)
    fields = [(n, re.sub(r"\s+", "", t)) for n, t in re.findall(r"pub (\w+)\s*:\s*([^,\n]+)", re.sub(r"//[^\n]*", "", m.group(1)))]
    assert fields == [("dtype", "u8"), ("pad_", "[u8;7]"), ("bits", "u64")], fields
    # repr(C): u8 at 0, seven bytes at 1, u64 aligned to 8 -> offset 8, size 16
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "erased_cells.h"\n'
                    'int main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(ec_value), offsetof(ec_value, dtype), '
                    'offsetof(ec_value, pad_), offsetof(ec_value, v), sizeof(ec_comm_uid)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [16, 0, 1, 8, 128]
    assert re.search(r"pub struct ec_comm_uid\s*\{\s*pub bytes:\s*\[c_char;\s*128\]", rs)


def test_ec_expr_step_layout_and_operand_references(tmp_path):
    """`ec_expr_step` is four `int8_t` in the order op, a, b, dst on both sides (4 bytes, no padding), and the operand
    reference helpers and limits of ffi.rs give the numbers of the header's macros and enum."""
    rs = open(FFI_RS).read()
    m = re.search(r"#\[repr\(C\)\]\s*#\[derive\([^)]*\)\]\s*pub struct ec_expr_step\s*\{([^}]*)\}", rs)
    assert m, "ffi.rs lacks #[repr(C)] ec_expr_step"
    fields = re.findall(r"pub (\w+)\s*:\s*(\w+)", m.group(1))
    assert fields == [("op", "i8"), ("a", "i8"), ("b", "i8"), ("dst", "i8")], fields
    prog = tmp_path / "step.c"
    prog.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "erased_cells.h"\n'
                    'int main(void) { printf("%zu %zu %zu %zu %zu %d %d %d %d %d %d %d\\n", sizeof(ec_expr_step), '
                    'offsetof(ec_expr_step, op), offsetof(ec_expr_step, a), offsetof(ec_expr_step, b), offsetof(ec_expr_step, dst), '
                    'EC_EXPR_STREAM(3), EC_EXPR_REG(3), EC_EXPR_SCALAR(7), EC_EXPR_MAX_STREAMS, EC_EXPR_REGS, '
                    'EC_EXPR_MAX_SCALARS, EC_EXPR_MAX_STEPS); return 0; }\n')
    exe = tmp_path / "step"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert out[:5] == [4, 0, 1, 2, 3]
    base = {k: int(v) for k, v in re.findall(r"pub const fn ec_expr_(stream|reg|scalar)\(k: i8\) -> i8 \{\s*(?:(\d+) \+ )?k\s*\}", rs)
            for v in [v or "0"]}
    assert [base["stream"] + 3, base["reg"] + 3, base["scalar"] + 7] == out[5:8]
    lim = {k: int(v) for k, v in re.findall(r"pub const (EC_EXPR_\w+): usize = (\d+);", rs)}
    assert [lim["EC_EXPR_MAX_STREAMS"], lim["EC_EXPR_REGS"], lim["EC_EXPR_MAX_SCALARS"], lim["EC_EXPR_MAX_STEPS"]] == out[8:]

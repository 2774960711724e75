"""The Rust crate (erased-cells_amd/rust/erased-cells-hip) cannot be compiled in this image (no rustc), so its
public surface is held in lockstep with the reference's by text: every `pub fn`, trait method, `pub
trait/enum/struct/type` and `impl ... for ...` header of the reference files on the path (src/lib.rs, buffer.rs,
value.rs, ctype.rs, encoding.rs, error.rs, masked/*.rs — unit-test modules excluded) must appear, same name and
same signature, in the crate.  Runs only where the reference tree is present (this container); on a box without
it the test is skipped.  (The crate itself does reproduce the reference's declarations — that is the point of a
drop-in; which lines, under which licence, is recorded in INTEGRATION.md §2 and held by tests/test_rust_provenance.py.)"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
CRATE = os.path.join(ROOT, "erased-cells_amd", "rust", "erased-cells-hip", "src")
REF_FILES = ["lib.rs", "buffer.rs", "value.rs", "ctype.rs", "encoding.rs", "error.rs", "masked/mask.rs",
             "masked/masked_buffer.rs", "masked/nodata.rs"]

# Items of the reference the device-resident crate deliberately does not carry, each with its reason.
DEVIATIONS = {
    "pubenumCellBuffer{$($id(Vec<$p>)),*}": "CellBuffer is a struct (cell-type tag + length + HBM block), not an enum over Vec<T>",
    "pubstructMask(Vec<bool>)": "Mask is a struct over an HBM block (plus an optional host shadow for &mut bool access)",
    "pubenumCellType{$($id),*}": "written out variant by variant with explicit discriminants (same names, same order)",
    "fnrender<T:Debug>(values:&[T],f:&mutFormatter<'_>)->std::fmt::Result": "private helper nested in `Debug for Elided` (not API)",
}


def _strip_tests_and_comments(text: str) -> str:
    i = text.find("#[cfg(test)]")
    if i >= 0:
        text = text[:i]
    text = re.sub(r"//[^\n]*", "", text)
    return text


def _items(text: str) -> set:
    """Normalised headers: whitespace removed, up to the opening `{` / `;` (tuple structs: up to `;`)."""
    text = _strip_tests_and_comments(text)
    out = set()
    # fn signatures (pub or trait methods / trait-impl methods)
    for m in re.finditer(r"(?:^|\n)\s*((?:pub\s+)?fn\s+[A-Za-z_][A-Za-z0-9_]*[^{;]*)[{;]", text):
        sig = m.group(1)
        if sig.lstrip().startswith("pub(crate)") or "$mth" in sig:
            continue
        out.add(_norm(sig))
    for m in re.finditer(r"(?:^|\n)\s*(impl\b[^{;]*)\{", text):
        out.add(_norm(m.group(1)))
    for m in re.finditer(r"(?:^|\n)\s*(pub\s+(?:trait|enum|struct|type)\s+[^{;]*)(\{[^}]*\}|;)", text):
        head, body = m.group(1), m.group(2)
        kind = head.split()[1]
        out.add(_norm(head + (body if kind == "enum" and "$" in body else "")))
    return out


def _norm(s: str) -> str:
    s = re.sub(r"\s+", "", s)
    s = s.replace("crate::error::Result", "Result").replace("error::Result", "Result")
    s = re.sub(r",(?=$|\))", "", s)       # trailing commas
    s = re.sub(r"^pubfn", "fn", s)        # a trait method and its inherent twin count as the same item
    return s


def _crate_items() -> set:
    items = set()
    for f in os.listdir(CRATE):
        if f.endswith(".rs"):
            items |= _items(open(os.path.join(CRATE, f)).read())
    return items


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present on this box")
def test_every_public_item_of_the_reference_exists_in_the_rust_crate():
    have = _crate_items()
    missing = {}
    for rel in REF_FILES:
        want = _items(open(os.path.join(REF, rel)).read())
        gone = sorted(w for w in want if w not in have and w not in DEVIATIONS)
        if gone:
            missing[rel] = gone
    assert not missing, "reference items without a same-signature item in the Rust crate:\n" + \
        "\n".join(f"  {k}: {v}" for k, v in missing.items())


def test_crate_is_named_like_the_reference_library_and_carries_its_macro():
    toml = open(os.path.join(os.path.dirname(CRATE), "Cargo.toml")).read()
    assert re.search(r'\[lib\]\s*name\s*=\s*"erased_cells"', toml), "`use erased_cells::...` must resolve to this crate"
    assert "num-traits" in toml  # CellEncoding: Zero + One, CellValue: ToPrimitive
    lib = open(os.path.join(CRATE, "lib.rs")).read()
    assert "#[macro_export]" in lib and "macro_rules! with_ct" in lib
    for name in ("UInt8, u8", "UInt16, u16", "UInt32, u32", "UInt64, u64", "Int8, i8", "Int16, i16", "Int32, i32",
                 "Int64, i64", "Float32, f32", "Float64, f64"):
        assert f"({name})" in lib
    # no foreign communicator in the sharded API any more
    sh = open(os.path.join(CRATE, "sharded.rs")).read()
    assert "rccl_comm: *mut c_void" not in sh and "ec_comm_init_rank" in sh and "ec_shard_group_create" in sh


def _strip_rust(text: str) -> str:
    """Rust source with comments, string / char literals and lifetimes blanked (enough of a lexer for the checks below)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            depth, i = 1, i + 2
            while i < n and depth:
                if text.startswith("/*", i):
                    depth, i = depth + 1, i + 2
                elif text.startswith("*/", i):
                    depth, i = depth - 1, i + 2
                else:
                    i += 1
        elif c == '"' or (c == "r" and re.match(r'r#*"', text[i:])):
            if c == "r":
                m = re.match(r'r(#*)"', text[i:])
                end = text.find('"' + m.group(1), i + len(m.group(0)))
                i = n if end < 0 else end + 1 + len(m.group(1))
            else:
                i += 1
                while i < n and text[i] != '"':
                    i += 2 if text[i] == "\\" else 1
                i += 1
            out.append('""')
        elif c == "'":
            m = re.match(r"'(\\.[^']*|[^'\\])'", text[i:])  # a char literal; anything else is a lifetime
            if m:
                i += len(m.group(0))
                out.append("' '")
            else:
                i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def test_rust_sources_are_balanced_and_call_only_declared_ffi_items():
    """No rustc here, so the cheapest classes of breakage are held by text: every bracket of every source file closes in
    order, and every `ec_*` name a module uses is declared in ffi.rs (a call to an entry point the header has but ffi.rs
    lacks — or a typo — would only show at the user's first `cargo build`)."""
    src = CRATE
    ffi = _strip_rust(open(os.path.join(src, "ffi.rs")).read())
    declared = set(re.findall(r"\b(?:fn|struct|type|const|static)\s+(ec_\w+|EC_\w+)", ffi))
    assert {"ec_binop", "ec_expr", "ec_value", "ec_expr_step", "EC_ADD", "ec_expr_reg"} <= declared
    pairs = {")": "(", "]": "[", "}": "{"}
    for name in sorted(os.listdir(src)):
        if not name.endswith(".rs"):
            continue
        text = _strip_rust(open(os.path.join(src, name)).read())
        stack = []
        for k, c in enumerate(text):
            if c in "([{":
                stack.append((c, k))
            elif c in ")]}":
                assert stack and stack[-1][0] == pairs[c], f"{name}: unbalanced '{c}' near: {text[max(0, k - 60):k + 20]!r}"
                stack.pop()
        assert not stack, f"{name}: unclosed '{stack[-1][0]}' near: {text[stack[-1][1]:stack[-1][1] + 60]!r}"
        if name != "ffi.rs":
            used = set(re.findall(r"\b(ec_[a-z0-9_]+|EC_[A-Z0-9_]+)\b", text))
            local = set(re.findall(r"\b(?:fn|let|mod|struct|type|const)\s+(ec_\w+|EC_\w+)", text))
            unknown = used - declared - local
            assert not unknown, f"{name} uses {sorted(unknown)}: not declared in ffi.rs"

"""No-GPU checks of the drop-in boundary: the shared library loads, exports every
symbol include/erased_cells.h declares, its host-side lattice agrees with the
oracle (and so with src/ctype.rs), and compute calls fail loudly — never fall
back — when no HIP device is bound."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

from oracle import eco

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ec():
    import erased_cells_hip as ec
    return ec


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "erased_cells.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef[^;{]*\(\*[^;]*;", "", text)  # function-pointer typedefs (ec_shard_fn) are not entry points
    return sorted(set(re.findall(r"\b(ec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ec):
    declared = _declared_symbols()
    assert len(declared) >= 40
    raw = C.CDLL(ec._ffi.SO_PATH)
    missing = [s for s in declared if not hasattr(raw, s)]
    assert not missing, f"declared in erased_cells.h but not exported: {missing}"
    # and the binding covers the whole header (no silently unbound entry point)
    assert sorted(ec._ffi.SIGNATURES) == declared
    assert ec.lib().ec_abi_version() == 1


def test_every_entry_point_cites_the_reference():
    """include/*.h must cite the reference interface each function replaces (file:line)."""
    text = open(os.path.join(ROOT, "include", "erased_cells.h")).read()
    for fn in ("ec_binop", "ec_binop_scalar", "ec_masked_binop", "ec_neg", "ec_convert", "ec_min_max",
               "ec_mask_from_nodata", "ec_mask_select", "ec_mask_and", "ec_mask_or", "ec_mask_not", "ec_mask_counts",
               "ec_union", "ec_can_fit_into"):
        i = text.index(fn + "(")
        ctx = text[max(0, i - 700): i + 300]
        assert re.search(r"src/[a-z_/]+\.rs:\d+", ctx), fn


def test_lattice_matches_oracle(ec):
    L = ec.lib()
    sizes = [1, 2, 4, 8, 1, 2, 4, 8, 4, 8]
    neg = {eco.U8: eco.I16, eco.U16: eco.I32, eco.U32: eco.F64, eco.U64: eco.F64}
    for a in range(eco.NTYPES):
        assert L.ec_size_of(a) == sizes[a]
        assert L.ec_neg_result_type(a) == neg.get(a, a)
        for b in range(eco.NTYPES):
            assert L.ec_union(a, b) == eco.union(a, b), (a, b)
            assert bool(L.ec_can_fit_into(a, b)) == eco.can_fit_into(a, b), (a, b)
        mn, mx, nd = ec._ffi.EcValue(), ec._ffi.EcValue(), ec._ffi.EcValue()
        assert L.ec_min_value(a, C.byref(mn)) == 0 and L.ec_max_value(a, C.byref(mx)) == 0
        assert ec.CellValue.from_ec(mn).bits() == eco.min_value(a).bits()
        assert ec.CellValue.from_ec(mx).bits() == eco.max_value(a).bits()
        assert L.ec_nodata_default(a, C.byref(nd)) == 0
        assert ec.CellValue.from_ec(nd).bits() == eco.nodata_value(eco.ND_DEFAULT, a).bits()


def test_value_convert_matches_oracle(ec):
    from vectors import rand_cells
    for s in range(eco.NTYPES):
        vals = rand_cells(s, 40, 5)
        for d in range(eco.NTYPES):
            for x in vals[:12]:
                if eco.can_fit_into(s, d):
                    got = ec.CellValue(s, x).convert(d)
                    exp = eco.value_convert(eco.Value.of(s, x), d)
                    assert (got.ct, got.bits()) == (exp.ct, exp.bits())
                    assert ec.CellValue(s, x).to_f64() == eco.value_to_f64(eco.Value.of(s, x)) or np.isnan(x)
                else:
                    with pytest.raises(ec.NarrowingError) as ei:
                        ec.CellValue(s, x).convert(d)
                    assert (ei.value.src, ei.value.dst) == (s, d)


def test_cell_value_ordering_matches_oracle(ec):
    from vectors import rand_cells
    for a_ct in range(eco.NTYPES):
        for b_ct in range(eco.NTYPES):
            for x, y in zip(rand_cells(a_ct, 6, 1), rand_cells(b_ct, 6, 2)):
                exp = eco.value_cmp(eco.Value.of(a_ct, x), eco.Value.of(b_ct, y))
                assert ec.CellValue(a_ct, x).cmp(ec.CellValue(b_ct, y)) == exp, (a_ct, x, b_ct, y)


def test_shard_range_row_blocks(ec):
    from erased_cells_hip import sharded
    # 169 rows over 8 shards: 22,21,...  (SURVEY §8d config 5); 16384 -> 2048 each
    rows = [sharded.shard_range(169, 186, g, 8) for g in range(8)]
    assert [ln // 186 for _, ln in rows] == [22, 21, 21, 21, 21, 21, 21, 21]
    assert rows[0][0] == 0 and all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(7))
    assert rows[-1][0] + rows[-1][1] == 169 * 186
    for g in range(8):
        assert sharded.shard_range(16384, 16384, g, 8) == (g * 2048 * 16384, 2048 * 16384)
    assert sharded.shard_range(3, 5, 3, 4) == (15, 0)  # more shards than rows: empty tail shards
    with pytest.raises(ec.EcError):
        sharded.shard_range(10, 10, 4, 4)


def test_min_max_key_decode_roundtrip(ec):
    """The all-reduce payload: {~key(min), key(max)} as int64, order-preserving for every type."""
    from erased_cells_hip import sharded
    from vectors import rand_cells
    for ct in range(eco.NTYPES):
        vals = rand_cells(ct, 200, 3)
        keys = [ec.CellValue(ct, v) for v in vals]
        srt = sorted(keys, key=lambda v: v._key(ct))
        for lo, hi in zip(srt, srt[1:]):
            assert eco.value_cmp(eco.Value.of(ct, lo.value), eco.Value.of(ct, hi.value)) <= 0
        # decode(encode(x)) == x, through the library's decode and the mirror's host-side key
        for v in keys[:50]:
            k = v._key(ct)
            if ct == eco.U64:
                k -= 1 << 63  # u64 keys are biased into int64
            mn, mx = sharded.combine_min_max_keys(ct, (~k, k))
            assert mn.bits() == v.bits() and mx.bits() == v.bits()


def test_no_cpu_fallback_without_device(ec):
    """On a box without a GPU every compute entry point must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(ec.EcError) as ei:
        ec.init(0)
    assert ei.value.status == ec._ffi.EC_ERR_HIP
    out = (C.c_double * 4)()
    a = (C.c_uint8 * 4)(1, 2, 3, 4)
    st = ec.lib().ec_binop(ec.DIV, ec.UInt8, a, ec.UInt8, a, 4, out, None)
    assert st == ec._ffi.EC_ERR_NOT_INITIALIZED
    assert b"ec_init" in ec.lib().ec_last_error_string()


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under erased-cells_amd/ or include/ may mention it."""
    bad = []
    for base in ("erased-cells_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in dp:
                continue
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"ec_oracle|libec_oracle|from oracle|import oracle|\beco\.", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_rust_extern_block_declares_every_entry_point():
    """The Rust binding (erased-cells_amd/rust/erased-cells-hip/src/ffi.rs) cannot be compiled in this image
    (no Rust toolchain), so its `extern "C"` block is at least kept in lockstep with the header: same set of
    symbols, same number of arguments each."""
    import re
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "erased_cells.h")).read(), flags=re.S)
    hdr = re.sub(r"typedef[^;{]*\(\*[^;]*;", "", hdr)
    c_decls = {m.group(1): m.group(2) for m in re.finditer(r"\b(ec_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr)}
    rs = re.sub(r"//.*", "", open(os.path.join(ROOT, "erased-cells_amd", "rust", "erased-cells-hip", "src", "ffi.rs")).read())
    r_decls = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (ec_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", rs, flags=re.S)}

    def argc(a):
        a = a.strip()
        return 0 if a in ("", "void") else len([x for x in a.split(",") if x.strip()])

    assert set(c_decls) == set(r_decls), (sorted(set(c_decls) - set(r_decls)), sorted(set(r_decls) - set(c_decls)))
    assert {k: argc(v) for k, v in c_decls.items()} == {k: argc(v) for k, v in r_decls.items()}


def _build_c_example(tmp_path, name="quick"):
    import subprocess
    exe = str(tmp_path / (name + "_c"))
    libdir = os.path.join(ROOT, "erased-cells_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", name + ".c"), "-L" + libdir, "-lerased_cells_hip",
                        "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c99_and_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: the header compiles as pedantic C99 and a C program links against the .so."""
    _build_c_example(tmp_path)
    _build_c_example(tmp_path, "evi")


@pytest.mark.gpu
def test_evi_expression_program_from_plain_c(tmp_path):
    """examples/evi.c: an eight-operator tree as one ec_expr call from C equals the eager chain of the same operators bit for
    bit — as the built-in kernel of the ahead-of-time catalogue, interpreted, and compiled for itself."""
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path, "evi")], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "one pass == eight passes; built-in launches 1, interpreted launches 1, compiled launches 1" in r.stdout


@pytest.mark.gpu
def test_quick_example_from_plain_c(tmp_path):
    """examples/quick.rs:5-11 through the C ABI from C (examples/quick.c)."""
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["0.25", "0.25", "0.25"]


def test_cell_value_scalar_semantics_reference_tests():
    """src/value.rs:283-360 restated on the Python mirror's CellValue (host-only: lattice functions of the library)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
    import erased_cells_hip as ec
    CV = ec.CellValue
    for ct in ec.CELL_TYPES:  # cell_type(), get()
        cv = CV(ct, 0)
        assert cv.cell_type() == ct and cv.get(ct) == 0 and cv.get(ec.Float64) == 0.0
    assert CV(ec.UInt8, 43).convert(ec.Int16) == CV(ec.Int16, 43) and CV(ec.UInt8, 43).convert(ec.Int16).cell_type() == ec.Int16
    with pytest.raises(ec.NarrowingError):
        CV(ec.Float32, 3.11111).convert(ec.Int32)
    assert CV(ec.Float32, 3.11111).convert(ec.Float32) == CV(ec.Float32, 3.11111)
    r = CV(ec.UInt16, 33).convert(ec.Float32)
    assert r.cell_type() == ec.Float32 and r.value == np.float32(33.0)
    assert CV.zero().is_zero() and not CV.one().is_zero()
    for (ct, v), (ect, ev) in [((ec.UInt8, 1), (ec.Int16, -1)), ((ec.UInt16, 1), (ec.Int32, -1)), ((ec.Int8, 1), (ec.Int8, -1)),
                               ((ec.Int16, 1), (ec.Int16, -1)), ((ec.Float64, 1.0), (ec.Float64, -1.0)),
                               ((ec.Float32, 1.0), (ec.Float32, -1.0)), ((ec.UInt32, 5), (ec.Float64, -5.0)),
                               ((ec.Int8, -128), (ec.Int8, -128))]:
        n = -CV(ct, v)
        assert n.cell_type() == ect and n.value == ev
    for ct in (ec.UInt8, ec.UInt16, ec.Int64, ec.Float32):  # binops: always Float64
        l, r = CV(ct, 1), CV(ct, 2)
        for got, exp in [(l + r, 3.0), (l + 2, 3.0), (l - r, -1.0), (l - 2, -1.0), (r - l, 1.0), (l * r, 2.0), (r * l, 2.0),
                         (l / r, 0.5), (r / l, 2.0)]:
            assert got.cell_type() == ec.Float64 and got == CV(ec.Float64, exp)
    a, b = CV(ec.UInt8, 3).unify(CV(ec.Int8, -3))
    assert a.cell_type() == b.cell_type() == ec.Int16 and (a.value, b.value) == (3, -3)
    assert CV(ec.UInt8, 3) < CV(ec.Float32, 3.5) and CV(ec.Int64, 4) > CV(ec.UInt8, 3)


def test_cell_type_reference_tests():
    """src/ctype.rs:185-290 restated on the Python mirror (can_union, is_integral, size, has_min_max, can_string, zero_one)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
    import erased_cells_hip as ec
    for ct in (ec.UInt8, ec.UInt16, ec.Float32, ec.Float64):
        assert ec.union(ct, ct) == ct
    assert ec.union(ec.Int16, ec.Float32) == ec.Float32 == ec.union(ec.Float32, ec.Int16)
    assert ec.union(ec.UInt8, ec.UInt16) == ec.UInt16 and ec.union(ec.Int32, ec.Float32) == ec.Float64
    assert ec.is_integral(ec.UInt8) and ec.is_integral(ec.UInt16) and not ec.is_integral(ec.Float32) and not ec.is_integral(ec.Float64)
    assert ec.is_signed(ec.Int8) and ec.is_signed(ec.Float32) and not ec.is_signed(ec.UInt64)
    assert [ec.size_of(ct) for ct in ec.CELL_TYPES] == [1, 2, 4, 8, 1, 2, 4, 8, 4, 8]
    for ct in ec.CELL_TYPES:
        dt = ec.NP_DTYPES[ct]
        lo, hi = (np.iinfo(dt).min, np.iinfo(dt).max) if dt.kind in "ui" else (np.finfo(dt).min, np.finfo(dt).max)
        assert ec.min_value(ct) == ec.CellValue(ct, lo) and ec.max_value(ct) == ec.CellValue(ct, hi)
        assert ec.cell_type_from_str(ec.cell_type_to_string(ct)) == ct
        assert ec.zero(ct).is_zero() and not ec.one(ct).is_zero() and ec.zero(ct).cell_type() == ct
    assert ec.cell_type_to_string(ec.UInt8) == "UInt8" and ec.cell_type_to_string(ec.Float64) == "Float64"
    with pytest.raises(ec.ParseError):
        ec.cell_type_from_str("UInt57")


def test_new_runtime_and_sharding_entry_points_reject_bad_arguments_without_a_device(ec):
    """Argument and state checks of the round-2 entry points that need no GPU: they fail with a status and a message,
    never crash, and nothing initialises a device behind the caller's back."""
    import torch
    L, E = ec.lib(), ec._ffi
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: the no-device statuses do not apply")
    dev = C.c_int32(-1)
    assert L.ec_set_device(0) == E.EC_ERR_NOT_INITIALIZED and b"ec_init" in L.ec_last_error_string()
    assert L.ec_get_device(C.byref(dev)) == E.EC_ERR_NOT_INITIALIZED
    assert L.ec_get_device(None) == E.EC_ERR_ARG
    assert L.ec_pool_trim(0) == E.EC_ERR_NOT_INITIALIZED
    assert L.ec_release_stream(None) == E.EC_ERR_NOT_INITIALIZED
    assert L.ec_free_ordered(None, None, None) == E.EC_OK           # freeing nothing is fine
    v = C.c_int64(-1)
    assert L.ec_stat_get(b"devices", C.byref(v)) == E.EC_OK and v.value == 0
    assert L.ec_stat_get(b"scratch_streams", C.byref(v)) == E.EC_OK and v.value == 0
    assert L.ec_stat_get(b"pool_allocs", C.byref(v)) == E.EC_OK and v.value == 0
    assert L.ec_stat_get(b"no such counter", C.byref(v)) == E.EC_ERR_ARG
    assert L.ec_stat_get(None, C.byref(v)) == E.EC_ERR_ARG
    for key in (b"pool_keep_mb", b"fused_mixed", b"reduce_shape", b"reduce_bpc"):
        assert L.ec_tune_set(key, 1) == E.EC_OK
    assert L.ec_tune_set(b"pool_keep_mb", 32768) == E.EC_OK and L.ec_tune_set(b"fused_mixed", 1) == E.EC_OK
    assert L.ec_tune_set(b"reduce_shape", 0) == E.EC_OK and L.ec_tune_set(b"reduce_bpc", 0) == E.EC_OK
    assert L.ec_tune_set(b"nonsense", 1) == E.EC_ERR_ARG
    # communicators: argument checks come before anything touches RCCL or a device
    uid, comm = E.EcCommUid(), C.c_void_p()
    assert L.ec_comm_get_unique_id(None) == E.EC_ERR_ARG
    assert L.ec_comm_init_rank(None, 1, 0, C.byref(comm)) == E.EC_ERR_ARG
    assert L.ec_comm_init_rank(C.byref(uid), 2, 2, C.byref(comm)) == E.EC_ERR_ARG
    assert L.ec_comm_init_rank(C.byref(uid), 1, 0, C.byref(comm)) == E.EC_ERR_NOT_INITIALIZED
    assert L.ec_comm_init_all(None, 1, None) == E.EC_ERR_ARG
    assert L.ec_comm_init_all((C.c_int32 * 2)(3, 3), 2, (C.c_void_p * 2)()) == E.EC_ERR_ARG and b"listed twice" in L.ec_last_error_string()
    assert L.ec_comm_destroy(None) == E.EC_OK
    assert L.ec_allreduce_min_max_keys(None, None, None) == E.EC_ERR_NOT_INITIALIZED
    # shard groups
    g = C.c_void_p()
    assert L.ec_shard_group_create(None, 1, 0, C.byref(g)) == E.EC_ERR_ARG
    assert L.ec_shard_group_create((C.c_int32 * 1)(0), 0, 0, C.byref(g)) == E.EC_ERR_ARG
    assert L.ec_shard_group_create((C.c_int32 * 2)(1, 1), 2, 0, C.byref(g)) == E.EC_ERR_ARG          # RCCL needs distinct devices
    assert L.ec_shard_group_create((C.c_int32 * 1)(0), 1, 1, C.byref(g)) == E.EC_ERR_HIP and not g.value  # no device here
    assert L.ec_shard_group_destroy(None) == E.EC_OK and L.ec_shard_group_size(None) == 0
    n1, p1 = (C.c_size_t * 1)(4), (C.c_void_p * 1)()
    for st in (L.ec_shard_group_sync(None), L.ec_sharded_alloc(None, n1, p1), L.ec_sharded_free(None, p1),
               L.ec_sharded_binop(None, 0, 0, p1, 0, p1, n1, p1), L.ec_sharded_convert(None, 0, p1, 1, p1, n1),
               L.ec_sharded_counts(None, p1, n1, C.byref(C.c_uint64()), C.byref(C.c_uint64())),
               L.ec_shard_group_shard(None, 0, None, None)):
        assert st == E.EC_ERR_ARG and b"null shard group" in L.ec_last_error_string()
    # still nothing initialised
    assert L.ec_stat_get(b"devices", C.byref(v)) == E.EC_OK and v.value == 0

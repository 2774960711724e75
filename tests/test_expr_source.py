"""The run-time compiled form of an expression program, as far as it goes without a GPU: `ec_expr_source` validates a
program, writes the HIP source the library would compile for it and — given a processor name — compiles it once with
hiprtc (which cross-compiles gfx950 here).  The cells the compiled kernels produce are checked on the GPU
(tests/test_gpu_instantiations.py::test_expr_compiled_form_equals_the_interpreter_and_the_oracle)."""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
import erased_cells_hip as ec  # noqa: E402

S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
EVI = [(ec.SUB, S(0), S(1), 0), (ec.MUL, R(0), K(0), 0), (ec.MUL, S(1), K(1), 1), (ec.ADD, S(0), R(1), 1),
       (ec.MUL, S(2), K(2), 2), (ec.SUB, R(1), R(2), 1), (ec.ADD, R(1), K(3), 1), (ec.DIV, R(0), R(1), 0)]
HAVE_HIPRTC = any(os.path.exists(os.path.join(d, "libhiprtc.so")) for d in ("/opt/rocm/lib", "/opt/rocm/lib64"))


def test_source_is_the_program_written_out():
    src = ec.fused.program_source([ec.UInt16, ec.Int8, ec.Float64], 4, EVI)
    assert 'extern "C" __global__ __launch_bounds__(256) void ec_expr_jit(' in src
    # the steps, in the program's order, on the operands the program names
    body = src[src.index("static __device__ __forceinline__ void run("):]
    want = ["r0[i] = s0[i] - s1[i];", "t[i] = r0[i] * c0;", "t[i] = s1[i] * c1;", "t[i] = s0[i] + r1[i];", "t[i] = s2[i] * c2;",
            "t[i] = r1[i] - r2[i];", "t[i] = r1[i] + c3;", "t[i] = r0[i] / r1[i];"]
    at = 0
    for line in want:
        at = body.index(line, at) + 1
    # every step keeps cv_bin_op!'s NaN rule — except the first: a difference of u16 and i8 cells is a finite integer, which the
    # generator can read off the program (and a divide of such values would be the short exact divide)
    assert body.count("NANFIX(") == len(EVI) - 1
    # typed loads: a pair of u16 cells is one 32-bit word, a pair of i8 cells one 16-bit word taken apart with shifts, f64 as is
    assert "(const W2*)b + pr" in src and "(int)(w << 24) >> 24" in src and "__builtin_bit_cast(double, x)" in src
    assert src.count("__builtin_nontemporal_load((const W") == 3, "every stream of the diagnostic source is non-temporal"
    # head offsets in bytes of each stream's cell type
    assert "p0 + (unsigned long)head * 2;" in src and "p1 + (unsigned long)head * 1;" in src and "p2 + (unsigned long)head * 8;" in src


def test_malformed_programs_are_refused_without_a_device():
    E, L = ec._ffi, ec.lib()
    dt = (C.c_uint8 * 1)(ec.UInt16)
    need = C.c_size_t(0)

    def run(steps, n_streams=1, n_scalars=1, dtv=dt):
        st = (E.EcExprStep * max(1, len(steps)))(*[E.EcExprStep(*q) for q in steps])
        return L.ec_expr_source(dtv, n_streams, n_scalars, st, len(steps), None, None, 0, C.byref(need))

    assert run([(ec.ADD, 0, 8, 0)]) == E.EC_OK and need.value > 1000
    assert run([(ec.ADD, 0, 4, 0)]) == E.EC_ERR_ARG       # register read before it is written
    assert run([(ec.ADD, 1, 8, 0)]) == E.EC_ERR_ARG       # no stream 1
    assert run([(ec.ADD, 0, 9, 0)]) == E.EC_ERR_ARG       # no scalar 1
    assert run([(ec.ADD, 0, 8, 4)]) == E.EC_ERR_ARG       # no register 4
    assert run([(5, 0, 8, 0)]) == E.EC_ERR_ARG
    assert run([]) == E.EC_ERR_ARG
    assert run([(ec.ADD, 0, 8, 0)] * 17) == E.EC_ERR_ARG
    assert run([(ec.ADD, 0, 8, 0)], dtv=(C.c_uint8 * 1)(11)) == E.EC_ERR_UNSUPPORTED_TYPE
    # a short buffer gets a truncated, terminated copy and the needed length
    buf = C.create_string_buffer(64)
    st = (E.EcExprStep * 1)(E.EcExprStep(ec.ADD, 0, 8, 0))
    assert L.ec_expr_source(dt, 1, 1, st, 1, None, buf, 64, C.byref(need)) == E.EC_OK
    assert len(buf.value) == 63 and need.value > 64


@pytest.mark.skipif(not HAVE_HIPRTC, reason="libhiprtc is not installed here")
@pytest.mark.timeout(300)
def test_generated_sources_compile_for_gfx950():
    """hiprtc accepts what the generator writes: EVI over three cell types, every cell type's loader, a 16-step program with
    all four ops and every operand kind on either side."""
    ec.fused.program_source([ec.UInt16, ec.Int8, ec.Float64], 4, EVI, arch="gfx950")
    ec.fused.program_source([ec.UInt8, ec.UInt32, ec.UInt64, ec.Int16], 1, [(ec.ADD, S(0), S(1), 0), (ec.DIV, S(2), S(3), 1), (ec.MUL, R(0), R(1), 2),
                                                                          (ec.SUB, K(0), R(2), 3)], arch="gfx950")
    ec.fused.program_source([ec.Int32, ec.Int64, ec.Float32], 0, [(ec.SUB, S(0), S(1), 0), (ec.DIV, R(0), S(2), 0)], arch="gfx950")
    long = [(k % 4, [S(0), R(0), K(k % 8)][k % 3] if k else S(0), [K((k + 3) % 8), S(0), R(0)][k % 3] if k else K(0), 0 if k % 5 else (1 if k else 0)) for k in range(16)]
    long = [(op, a if (a != R(0) or k > 0) else S(0), b if (b != R(0) or k > 0) else S(0), d) for k, (op, a, b, d) in enumerate(long)]
    ec.fused.program_source([ec.Float64], 8, long, arch="gfx950")
    with pytest.raises(Exception, match="targets gfx950"):
        ec.fused.program_source([ec.UInt16], 1, [(ec.ADD, S(0), K(0), 0)], arch="gfx000-no-such-processor")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc is not installed here")
@pytest.mark.timeout(300)
def test_generated_kernel_keeps_the_load_policy_in_the_machine_code(tmp_path):
    """The generated kernel in gfx950 assembly (hipcc, the flags the library gives hiprtc): every global load and store of the
    diagnostic source carries `nt` — the 1-byte streams included, which travel as 16-bit words because hipcc drops the flag
    from <N x i8> loads — no scratch, and no vector-memory access inside the steps."""
    import re
    import subprocess
    src = ec.fused.program_source([ec.UInt8, ec.Int8, ec.UInt16, ec.Float64], 4, [(ec.SUB, S(0), S(1), 0), (ec.MUL, R(0), K(0), 0), (ec.ADD, S(2), R(0), 1),
                                                                               (ec.DIV, R(1), S(3), 0), (ec.ADD, R(0), K(3), 2)])
    f = tmp_path / "k.hip"
    f.write_text("#include <hip/hip_runtime.h>\n" + src)
    asm = tmp_path / "k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only",
                    "-S", "-o", str(asm), str(f)], check=True, capture_output=True)
    text = asm.read_text()
    body = text[text.index("\nec_expr_jit:"):]
    body = body[:body.index("s_endpgm")]
    mem = [l.strip() for l in body.split("\n") if re.match(r"\s*global_(load|store)", l)]
    assert len(mem) >= 10, mem
    assert all(l.endswith(" nt") for l in mem), [l for l in mem if not l.endswith(" nt")]
    assert any("global_load_ushort" in l for l in mem), "a pair of 1-byte cells is one 16-bit word"
    assert "scratch_" not in body and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", text)


@pytest.mark.skipif(not HAVE_HIPRTC, reason="libhiprtc is not installed here")
@pytest.mark.timeout(300)
def test_reduce_variant_compiles_and_stores_nothing_but_its_keys(monkeypatch):
    """The variant `ec_expr_min_max` compiles (EC_EXPR_SOURCE_VARIANT=reduce): hiprtc accepts it; it loads the streams, folds order
    keys and ends in two atomic max — no store of cells; NDVI over u16 bands uses the short exact divide and no NaN tests."""
    monkeypatch.setenv("EC_EXPR_SOURCE_VARIANT", "reduce")
    src = ec.fused.program_source([ec.UInt16, ec.Int8, ec.Float64], 4, EVI, arch="gfx950")
    assert src.count("__hip_atomic_fetch_max(keys2") == 2 and "__builtin_nontemporal_store" not in src
    assert "for (unsigned long tile = blockIdx.x; tile < ntiles; tile += gridDim.x)" in src  # every wave reaches the end of its loop
    ndvi = [(ec.SUB, S(0), S(1), 0), (ec.ADD, S(0), S(1), 1), (ec.DIV, R(0), R(1), 0)]
    src = ec.fused.program_source([ec.UInt16, ec.UInt16], 0, ndvi, arch="gfx950")
    body = src[src.index("static __device__ __forceinline__ void run("):src.index("static __device__ __forceinline__ long long okey")]
    assert "= divs(r0[i], r1[i]);" in body and "NANFIX(" not in body


def _tree(cts, nscal, steps):
    src = ec.fused.program_source(cts, nscal, steps)
    lines = src.splitlines()
    assert lines[0].startswith("// tree: ") and lines[1].startswith("// ahead-of-time kernel: ")
    return lines[0][len("// tree: "):], lines[1][len("// ahead-of-time kernel: "):]


def test_a_program_is_recognised_by_its_tree_not_by_its_step_list():
    """The ahead-of-time catalogue (csrc/ec_expr_fixed.hpp) is looked up by the expression a step list computes: register
    names, the schedule of independent sub-trees, the numbering of streams and scalars and the side a lone scalar of + or *
    stands on do not matter; operand order of everything else does."""
    U16, I16, F32 = ec.UInt16, ec.Int16, ec.Float32
    evi_tree = "(/ (* (- S0 S1) K0) (+ (- (+ S0 (* S1 K1)) (* S2 K2)) K3))"
    assert _tree([U16, I16, U16], 4, EVI) == (evi_tree, "EVI")
    # the denominator first, other registers, scalars on the left, streams and scalars numbered differently:
    # 2.5 * (nir - red) / (nir + 6 * red - 7.5 * blue + 1) with nir = stream 2, red = stream 0, blue = stream 1
    other = [(ec.MUL, K(3), S(0), 3), (ec.ADD, S(2), R(3), 3), (ec.MUL, K(0), S(1), 0), (ec.SUB, R(3), R(0), 2), (ec.ADD, R(2), K(1), 2),
             (ec.SUB, S(2), S(0), 1), (ec.MUL, K(2), R(1), 1), (ec.DIV, R(1), R(2), 0)]
    assert _tree([U16, U16, U16], 4, other) == (evi_tree, "EVI")
    # NDVI with its sum first; a dead step in between drops out
    ndvi = [(ec.ADD, S(0), S(1), 2), (ec.MUL, S(0), S(0), 3), (ec.SUB, S(0), S(1), 1), (ec.DIV, R(1), R(2), 0)]
    assert _tree([F32, F32], 0, ndvi) == ("(/ (- S0 S1) (+ S0 S1))", "NDVI")
    # (red - nir) / (nir + red) is another tree: operand order of stream operands is kept
    assert _tree([F32, F32], 0, [(ec.SUB, S(1), S(0), 0), (ec.ADD, S(0), S(1), 1), (ec.DIV, R(0), R(1), 0)])[0] == "(/ (- S0 S1) (+ S1 S0))"
    assert _tree([ec.UInt8, ec.Int8, ec.UInt8], 0, [(ec.ADD, S(0), S(1), 0), (ec.MUL, R(0), S(2), 0)]) == ("(* (+ S0 S1) S2)", "add-mul")
    assert _tree([ec.Float64], 2, [(ec.MUL, K(1), S(0), 2), (ec.ADD, K(0), R(2), 2)]) == ("(+ (* S0 K0) K1)", "affine")
    # in the catalogue, but streams of different widths: the interpreter / the compiled form serve it
    t, k = _tree([U16, F32], 0, [(ec.SUB, S(0), S(1), 0), (ec.ADD, S(0), S(1), 1), (ec.DIV, R(0), R(1), 0)])
    assert t == "(/ (- S0 S1) (+ S0 S1))" and k.startswith("NDVI in the catalogue, but")
    # not in the catalogue; a subtraction's scalar is never moved; a register read twice is written out twice
    assert _tree([U16], 1, [(ec.SUB, K(0), S(0), 0)]) == ("(- K0 S0)", "none (not in the catalogue)")
    assert _tree([U16], 0, [(ec.ADD, S(0), S(0), 0), (ec.MUL, R(0), R(0), 1)])[0] == "(* (+ S0 S0) (+ S0 S0))"
    # a long chain has no bounded tree
    assert _tree([U16], 1, [(ec.MUL, S(0), K(0), 0)] + [(ec.ADD, R(0), R(0), 0)] * 15)[0].startswith("(none")

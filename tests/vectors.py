"""Seeded input generators shared by the oracle tests and the GPU parity tests."""
import numpy as np

from oracle.eco import NP_DTYPES

_F32_SPECIALS = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 3.4028235e38, -3.4028235e38,
                 1e-45, -1e-45, 1.17549435e-38, 16777216.0, 16777217.0, 0.1, 1e30, -1e30]
_F64_SPECIALS = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1.7976931348623157e308,
                 -1.7976931348623157e308, 5e-324, -5e-324, 2.2250738585072014e-308,
                 9007199254740992.0, 9007199254740993.0, 0.1, 1e300, -1e300, 1e-300]


def rand_cells(ct: int, n: int, seed: int, specials: bool = True) -> np.ndarray:
    """n cells of type ct: uniform over the whole range, with the type's extremes
    (and, for floats, signed zeros / inf / NaNs incl. payloads / subnormals) salted in."""
    rng = np.random.default_rng(seed * 1000003 + ct)
    dt = np.dtype(NP_DTYPES[ct])
    if dt.kind in "ui":
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
        if n and specials:
            sp = np.array([info.min, info.max, 0, 1, info.max - 1, info.min + 1, info.max // 2 + 1], dtype=dt)
            if dt.itemsize == 8:  # values around 2^53 where `as f64` rounds
                ext = [2**53, 2**53 + 1, 2**53 + 3, 2**62 + 1, 2**63 - 1025, 2**63 - 1]
                sp = np.concatenate([sp, np.array(ext, dtype=np.uint64).astype(dt)])
            idx = rng.integers(0, n, size=min(n, 4 * len(sp)))
            a[idx] = sp[np.arange(idx.size) % len(sp)]
        return a
    # floats: random bit patterns scaled to a sane range, plus specials
    if dt.itemsize == 4:
        a = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, size=n)).astype(np.float32)
        sp = np.array(_F32_SPECIALS, dtype=np.float32)
        nan_payloads = np.array([0x7FC00001, 0xFFC00000, 0x7F800001, 0xFFFFFFFF], dtype=np.uint32).view(np.float32)
    else:
        a = rng.standard_normal(n) * 10.0 ** rng.integers(-30, 31, size=n)
        sp = np.array(_F64_SPECIALS, dtype=np.float64)
        nan_payloads = np.array([0x7FF8000000000001, 0xFFF8000000000000, 0x7FF0000000000001,
                                 0xFFFFFFFFFFFFFFFF], dtype=np.uint64).view(np.float64)
    if n and specials:
        sp = np.concatenate([sp, nan_payloads])
        idx = rng.integers(0, n, size=min(n, 3 * len(sp)))
        a[idx] = sp[np.arange(idx.size) % len(sp)]
    return a


def rand_mask(n: int, seed: int, p_valid: float = 0.7) -> np.ndarray:
    rng = np.random.default_rng(seed ^ 0xA5A5)
    return (rng.random(n) < p_valid).astype(np.uint8)


def bits_of(a: np.ndarray) -> np.ndarray:
    """View as unsigned integers of the same width (bit-exact comparisons)."""
    return np.ascontiguousarray(a).view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def assert_f64_bits_equal(got: np.ndarray, exp: np.ndarray, nan_by_class_where=None):
    """Bit-exact f64 comparison. `nan_by_class_where`: boolean index of cells whose
    NaN sign/payload is not specified (both-NaN operands of a commutative op)."""
    g, e = bits_of(got), bits_of(exp)
    bad = g != e
    if nan_by_class_where is not None:
        both_nan = np.isnan(got) & np.isnan(exp) & nan_by_class_where
        bad &= ~both_nan
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        raise AssertionError(f"{int(bad.sum())} cells differ; first at {i}: got {got[i]!r} ({int(g[i]):#x}) "
                             f"expected {exp[i]!r} ({int(e[i]):#x})")


def both_nan(l: np.ndarray, r: np.ndarray) -> np.ndarray:
    with np.errstate(all="ignore"):
        return np.isnan(np.asarray(l).astype(np.float64)) & np.isnan(np.asarray(r).astype(np.float64))


def chain_loose(o1, x, y, o2, e1, e2, o3=None, z=None, w=None) -> np.ndarray:
    """Cells of `(x o1 y) o2 (z o3 w)` whose NaN sign/payload the reference does not pin — the single-op rule
    (both operands NaN under a commutative op: x86 returns whichever operand the compiler put first), carried
    through the chain: a step is loose where its own operands are both NaN under Add/Mul, or where an input of it
    was already loose (its payload is what propagates).  Everything else, NaNs included, must match bit for bit."""
    comm = (0, 2)  # ADD, MUL
    loose1 = both_nan(x, y) if o1 in comm else np.zeros(len(e1), bool)
    loose3 = (both_nan(z, w) if o3 in comm else np.zeros(len(e1), bool)) if o3 is not None else np.zeros(len(e1), bool)
    loose2 = both_nan(e1, e2) if o2 in comm else np.zeros(len(e1), bool)
    n = len(e1)
    return loose1[:n] | loose3[:n] | loose2

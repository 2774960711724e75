"""The round's new entry points from six host threads at once, for a few seconds: resident programs with the run mode changing
(interpreted / compiled at once / compiled in the background), host-to-host calls in the one-chunk form and through the
pipeline over the SAME numpy arrays, program statistics, and a shard group running programs and the sharded host pipeline —
every result compared with answers computed beforehand.  This is the test that found that two hiprtc compiles must not run at
once and that a page-lock registration must not come or go under another call's pageable copy of the same array
(EC_SOAK_SECONDS for longer runs).
"""
import os
import sys
import threading
import time
import traceback

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_new_entry_points_from_six_threads_at_once():
    import erased_cells_hip as ec
    from erased_cells_hip import sharded
    ec.init(0)
    P = ec.fused
    S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
    rng = np.random.default_rng(3)
    n = 1 << 20
    a = rng.integers(1, 40000, n).astype(np.uint16); b = rng.integers(1, 30000, n).astype(np.uint16); c = rng.uniform(-5, 5, n).astype(np.float32)
    da, db, dc = (ec.CellBuffer.from_vec(x) for x in (a, b, c))
    progs = [[(ec.SUB, S(0), S(1), 0), (ec.ADD, S(0), S(1), 1), (ec.DIV, R(0), R(1), 0)],
             [(ec.MUL, S(0), K(0), 0), (ec.ADD, R(0), S(2), 1), (ec.DIV, R(1), S(1), 0), (ec.SUB, R(0), K(1), 2)],
             [(ec.ADD, S(2), S(2), 0), (ec.MUL, R(0), S(0), 0)]]
    ec.lib().ec_tune_set(b"expr_jit", 0)
    want = [P.program([da, db, dc], [2.5, 0.25], pr).to_numpy() for pr in progs]
    want_mm = [P.program([da, db, dc], [2.5, 0.25], pr).min_max() for pr in progs]
    ec.lib().ec_tune_set(b"expr_jit", 1)
    stop = time.time() + float(os.environ.get("EC_SOAK_SECONDS", "8"))
    errors, counts = [], [0] * 6
    def guard(k, fn):
        def run():
            try:
                while time.time() < stop:
                    fn(); counts[k] += 1
            except BaseException as e:  # noqa: BLE001
                errors.append((k, repr(e)))
                print("thread", k, "failed after", counts[k], "iterations:", repr(e), flush=True)
                traceback.print_exc()
        return run
    def t_resident():
        i = counts[0] % 3
        ec.lib().ec_tune_set(b"expr_jit", [0, 2, 1][counts[0] % 3])
        assert np.array_equal(P.program([da, db, dc], [2.5, 0.25], progs[i]).to_numpy().view(np.uint64), want[i].view(np.uint64))
    def t_host_small():
        i = counts[1] % 3
        assert np.array_equal(P.program_host([a, b, c], [2.5, 0.25], progs[i]).view(np.uint64), want[i].view(np.uint64))
    def t_host_pipe():
        i = counts[2] % 3
        assert np.array_equal(P.program_host([a, b, c], [2.5, 0.25], progs[i], chunk_cells=1 << 17).view(np.uint64), want[i].view(np.uint64))
    def t_minmax():
        i = counts[3] % 3
        mn, mx = P.program_min_max([da, db, dc], [2.5, 0.25], progs[i])
        assert (mn.bits(), mx.bits()) == (want_mm[i][0].bits(), want_mm[i][1].bits())
    g = sharded.ShardGroup([0, 0, 0], host_combine=True)
    rows, cols = 1024, 1024
    sa, sb, sc_ = g.scatter(a, rows, cols), g.scatter(b, rows, cols), g.scatter(c, rows, cols)
    def t_group():
        i = counts[4] % 3
        out = g.program([sa, sb, sc_], [2.5, 0.25], progs[i])
        got = g.gather(out); out.free()
        assert np.array_equal(got.view(np.uint64), want[i].view(np.uint64))
        mn, mx = g.program_min_max([sa, sb, sc_], [2.5, 0.25], progs[i])
        assert (mn.bits(), mx.bits()) == (want_mm[i][0].bits(), want_mm[i][1].bits())
    def t_group_host():
        i = counts[4] % 3
        got = g.program_host([a, b, c], [2.5, 0.25], progs[i], rows, cols, chunk_cells=1 << 16)
        assert np.array_equal(got.view(np.uint64), want[i].view(np.uint64))
    def t_from_vec():  # the plain hand-over (pageable copies of the same arrays the pipelines page-lock)
        i = counts[5] % 3
        got = P.program([ec.CellBuffer.from_vec(a), ec.CellBuffer.from_vec(b), ec.CellBuffer.from_vec(c)], [2.5, 0.25], progs[i]).to_numpy()
        assert np.array_equal(got.view(np.uint64), want[i].view(np.uint64))
    fns = [t_resident, t_host_small, t_host_pipe, t_minmax, t_group, t_group_host]
    # the two group users share the group's call order; run them from ONE thread alternately
    def t_groups():
        t_group(); t_group_host()
    threads = [threading.Thread(target=guard(k, f)) for k, f in enumerate([t_resident, t_host_small, t_host_pipe, t_minmax, t_groups, t_from_vec])]
    for t in threads: t.start()
    for t in threads: t.join()


    g.__exit__(None, None, None)
    ec.lib().ec_tune_set(b"expr_jit", 1)
    assert not errors, errors[:3]
    assert all(c > 0 for c in counts), counts


def test_resident_api_from_six_threads_sharing_the_default_stream():
    """Operators, fused chains, masked operators, reductions (the per-stream scratch and its turn-taking), converts and device-side
    comparisons from six host threads on the library's default stream, each thread on its own buffers, every result checked."""
    import erased_cells_hip as ec
    ec.init(0)
    rng = np.random.default_rng(5)
    n = 300001
    stop = time.time() + float(os.environ.get("EC_SOAK_SECONDS", "6"))
    errors, counts = [], [0] * 6
    def worker(k):
        try:
            r = np.random.default_rng(100 + k)
            a = r.integers(0, 60000, n).astype(np.uint16); b = r.integers(1, 255, n).astype(np.uint8); c = r.uniform(-9, 9, n).astype(np.float32)
            m = r.integers(0, 2, n).astype(np.uint8)
            da, db, dc = ec.CellBuffer.from_vec(a), ec.CellBuffer.from_vec(b), ec.CellBuffer.from_vec(c)
            dm = ec.Mask.new(m)
            w_div = a.astype(np.float64) / b.astype(np.float64)
            w_chain = (a.astype(np.float64) + c.astype(np.float64)) * b.astype(np.float64)
            while time.time() < stop:
                q = da / db
                assert np.array_equal(q.to_numpy(), w_div)
                mn, mx = q.min_max()
                assert float(mn.value) == w_div.min() and float(mx.value) == w_div.max()
                f = ec.fused.expr(da, ec.ADD, dc, ec.MUL, db)
                assert np.array_equal(f.to_numpy(), w_chain)
                mq = ec.MaskedCellBuffer(da, dm) - ec.MaskedCellBuffer(db, dm)
                assert mq.counts() == (int(m.sum()), int(n - m.sum()))
                mmn, mmx = mq.min_max()
                v = (a.astype(np.float64) - b.astype(np.float64))[m.astype(bool)]
                assert float(mmn.value) == v.min() and float(mmx.value) == v.max()
                assert (da.convert(ec.Float32) == ec.CellBuffer.from_vec(a.astype(np.float32)))
                assert q == ec.CellBuffer.from_vec(w_div) and not (q == f)
                counts[k] += 1
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e))); print("thread", k, "failed:", repr(e), flush=True); traceback.print_exc()
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads: t.start()
    for t in threads: t.join()

    assert not errors, errors[:3]
    assert all(c > 0 for c in counts), counts


def test_shard_groups_from_three_threads():
    """Two shard groups on one device, one of them shared by two host threads: scatter, fire-and-forget operators, syncs, reductions,
    gathers and frees interleave; every answer checked (the group keeps each thread's calls in order and whole)."""
    import erased_cells_hip as ec
    from erased_cells_hip import sharded
    ec.init(0)
    rows, cols = 640, 480
    n = rows * cols
    stop = time.time() + float(os.environ.get("EC_SOAK_SECONDS", "5"))
    errors, counts = [], [0] * 3
    gA = sharded.ShardGroup([0] * 4, host_combine=True)
    gB = sharded.ShardGroup([0] * 3, host_combine=True)
    def worker(k, g):
        try:
            r = np.random.default_rng(200 + k)
            a = r.integers(1, 60000, n).astype(np.uint16); b = r.integers(1, 60000, n).astype(np.uint16)
            want = a.astype(np.float64) / b.astype(np.float64)
            while time.time() < stop:
                sa, sb = g.scatter(a, rows, cols), g.scatter(b, rows, cols)
                q = g.binop(ec.DIV, sa, sb)
                mn, mx = g.min_max(q)
                assert float(mn.value) == want.min() and float(mx.value) == want.max()
                assert np.array_equal(g.gather(q), want)
                s2 = g.binop(ec.SUB, q, q)
                g.sync()
                z0, z1 = g.min_max(s2)
                assert float(z0.value) == 0.0 and float(z1.value) == 0.0
                for x in (sa, sb, q, s2): x.free()
                counts[k] += 1
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e))); print("thread", k, "failed:", repr(e), flush=True); traceback.print_exc()
    threads = [threading.Thread(target=worker, args=(0, gA)), threading.Thread(target=worker, args=(1, gA)), threading.Thread(target=worker, args=(2, gB))]
    for t in threads: t.start()
    for t in threads: t.join()
    gA.__exit__(None, None, None); gB.__exit__(None, None, None)

    assert not errors, errors[:3]
    assert all(c > 0 for c in counts), counts

#!/usr/bin/env python3
"""bench.py — headline benchmark of the erased-cells MI355X path.

Workload (BASELINE.json configs[1]): 16384 x 16384 u8 ÷ u16 -> f64 `CellBuffer`
divide (src/buffer.rs:324-329 of the reference), inputs resident in HBM, one
"step" = one pass of the divide over the whole raster.  With --gpus N the raster
is cut into N contiguous row-blocks (one process / GPU); the divide
needs no data-path collective.  Scaling is therefore "strong": the raster is
fixed at 16384² (north_star) and each rank owns rows/N of it.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N --steps K --warmup W          (launches its own N ranks, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Timing: W warm-up steps, barrier + synchronize, K launches, synchronize, barrier.  Each rank's clock runs
from the opening barrier to its own closing synchronize (its K steps complete on the device); the MAX over
ranks is reported, so the closing barrier's own latency is outside the figure and a straggler is inside it.

The steps ROTATE over several operand sets (round 4): step i runs over set i mod S (S >= 4; each set has its
own operands and its own output, every rank its own sets), so that by the time a set's turn comes again at
least 1 GiB of other bytes has passed and nothing of it is left in the 256 MiB Infinity Cache.  `value`,
`ms_per_step` and `roofline.frac` are therefore ALL-HBM figures at every N: every algorithmic byte of a timed
step crosses HBM.  The loop of rounds 1-3 — one set, K steps, its 256 MiB u8 operand (at N = 8: both operands
of a shard) served on-die from the second step on — is measured after the timed region on every rank and
reported as `roofline.cache_resident_loop` (and per rank in `roofline.per_gpu[*].cache_resident_loop`).

Rank 0 prints ONE JSON line: BASELINE.json's metric (Gcells/s, whole job), a
`roofline` object for the dominant kernel (HIP-event timed, algorithmic bytes =
11 B/cell) and, at N=1, a `cpu_baseline` object: the oracle's reference-shaped
port of the same operation timed on this box's host cores on a bounded sample.

Other workloads (not bench lines; used for profiles/ and DESIGN.md):
  --workload masked_chain   config 3: f32 (a+b)*c with 30 % nodata masks
  --workload minmax         config 4 per-GPU shard: u16 min_max (+ all-reduce of the keys)
  --workload ndvi           config 5's arithmetic at raster scale: eager, --fused (one pass), --mixed (u16 + f32 bands)
  --workload evi            an eight-operator tree over three u16 bands: eager, --fused (ec_expr, the program compiled for
                            itself), --fused --interpret (the interpreter kernel)
  --workload binop          any cell-type pair and operator (--lt --rt --op)
  --e2e                     adds the host-memory-in / host-memory-out legs (naive from_vec/to_vec; ec_host_expr)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
METRIC = "Gcells/s + HBM-GB/s roofline %, u8/u16->f64 16384^2, 1/2/4/8 GPUs"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--side", type=int, default=16384, help="raster is side x side cells")
    ap.add_argument("--rows", type=int, default=None,
                    help="raster rows (default: side); e.g. --rows 2048 times on one GPU the row-block one rank owns at N=8")
    ap.add_argument("--workload", default="div_u8_u16", choices=["div_u8_u16", "masked_chain", "minmax", "ndvi", "binop", "evi"])
    ap.add_argument("--lt", default="u16", help="--workload binop: lhs cell type (u8 u16 u32 u64 i8 i16 i32 i64 f32 f64)")
    ap.add_argument("--rt", default="u16", help="--workload binop: rhs cell type")
    ap.add_argument("--interpret", action="store_true", help="--workload evi --fused: the interpreter kernel (k_expr; expr_fixed = 0, expr_jit = 0) "
                    "instead of the built-in straight-line kernel of the ahead-of-time catalogue (the library's default for this program)")
    ap.add_argument("--compiled", action="store_true", help="--workload evi --fused: the program compiled for itself with hiprtc (expr_fixed = 0, "
                    "expr_jit = 2): what the library does for programs outside its catalogue once they have run long enough")
    ap.add_argument("--op", default="add", choices=["add", "sub", "mul", "div"], help="--workload binop: operator")
    ap.add_argument("--fused", action="store_true", help="masked_chain / ndvi: the single-pass fused kernel instead of the eager chain")
    ap.add_argument("--mixed", action="store_true", help="ndvi: red band as f32 (mixed operand types -> the generic fused kernel)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: one side² raster row-sharded over the ranks (default); weak: side² per rank")
    ap.add_argument("--variant", type=int, default=None, help="binop kernel variant: 0 direct, 1 LDS-staged")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; nccl (= RCCL over xGMI) is the product path, gloo only for rehearsals")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (implies nothing about scaling)")
    ap.add_argument("--ramp", type=int, default=120, help="minimum untimed clock-ramp launches before the warm-up (0 = none)")
    ap.add_argument("--ramp-ms", type=float, default=40.0, help="minimum wall time of the clock ramp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", action="store_true",
                    help="also time from_vec (H2D) + divide + to_vec (D2H) once; reported beside, never as `value`")
    ap.add_argument("--graph", action="store_true",
                    help="capture the K timed steps into one hipGraph and replay it (every launch still executes); "
                         "removes the host's per-launch cost from short steps such as a 1/8 shard's ≈55 µs divide")
    ap.add_argument("--sets", type=int, default=0,
                    help="operand sets the steps rotate over (default: 4, more for small shards so that >= 1 GiB of other "
                         "sets passes between two uses of a byte); 1 = the one-set loop of rounds 1-3 as the timed region")
    ap.add_argument("--no-resident-loop", "--no-fresh-inputs", dest="no_resident_loop", action="store_true",
                    help="skip the untimed one-set loop measured after the timed region (roofline.cache_resident_loop)")
    ap.add_argument("--no-reference-streams", action="store_true",
                    help="skip the untimed reference streams (roofline.reference_streams) measured after the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="ec_tune_set(KEY, VALUE) before anything is launched (A/B runs for profiles/, e.g. --tune mall_mb=0); "
                         "recorded in config.tune")
    return ap.parse_args()


def cpu_baseline(side: int, target_s: float) -> dict:
    """The oracle (test infrastructure) as the reported CPU baseline: the reference-shaped port of
    src/buffer.rs:327 + src/value.rs:199-217 + src/buffer.rs:229-250, single thread like the reference."""
    import numpy as np
    from oracle import eco

    probe = 1 << 20
    a, b = eco.fill_u8(probe, 0x5EED0001), eco.fill_u16(probe, 0x5EED0002, lo=1)
    eco.binop(eco.DIV, a, b)  # warm
    t = time.perf_counter()
    eco.binop(eco.DIV, a, b)
    rate = probe / (time.perf_counter() - t)
    rows = max(8, min(side, int(rate * target_s) // side))
    n = rows * side
    a, b = eco.fill_u8(n, 0x5EED0001), eco.fill_u16(n, 0x5EED0002, lo=1)
    t = time.perf_counter()
    out = eco.binop(eco.DIV, a, b)
    dt = time.perf_counter() - t
    res = {"value": n / dt / 1e9, "unit": "Gcells/s", "cores": 1, "kind": "port",
           "sample": f"first {rows} rows x {side} cols of the same raster ({n} cells, {dt:.1f} s), "
                     f"reference-shaped oracle (16-byte tagged cells, per-cell union/convert, two-pass collect)"}
    # next to it: the typed-loop form on all host cores ("optimised CPU")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)
    eco.set_threads(cores)
    o2 = np.empty(n, dtype=np.float64)
    eco.f_binop(eco.DIV, a, b, o2)
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        eco.f_binop(eco.DIV, a, b, o2)
    dt2 = (time.perf_counter() - t) / reps
    eco.set_threads(0)
    res["typed_loop_all_cores"] = {"value": n / dt2 / 1e9, "unit": "Gcells/s", "cores": cores}
    assert np.array_equal(out, o2)
    return res


def e2e_pipelined(torch, ec, L, a, b, out, n: int, chunk: int = 1 << 25) -> dict:
    """Host-resident operands -> divide -> host-resident result through the library's host-to-host pipeline
    (`ec_host_expr`, csrc/ec_hostpipe.hip): page-locked host buffers, chunks of 2^25 cells, double-buffered device staging,
    one HIP stream each for H2D, the kernel and D2H, ordered by events.  Never `value`: this is what PCIe allows when the
    data must start and end on the host (8 B/cell come back), next to the naive pageable from_vec/to_vec path."""
    import numpy as np
    chk = ec._ffi.check
    P = ec.fused
    ha, hb, ho = P.pinned_empty(n, np.uint8), P.pinned_empty(n, np.uint16), P.pinned_empty(n, np.float64)
    # the benchmark's own inputs, copied to the page-locked host buffers once (untimed)
    chk(L.ec_download(C.c_void_p(ha.ctypes.data), a.mem.ptr, n, None))
    chk(L.ec_download(C.c_void_p(hb.ctypes.data), b.mem.ptr, 2 * n, None))
    prog = [(ec.DIV, 0, 1, 0)]  # a single operator is a one-step program

    def run():
        P.program_host([ha, hb], [], prog, out=ho, chunk_cells=chunk)

    run()  # warm (page tables of the pinned buffers, the pool's staging blocks)
    best = None
    for _ in range(3):
        t = time.perf_counter()
        run()
        dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    dt = best
    nchunks = (n + chunk - 1) // chunk
    # the pipelined result is the resident result: compare two chunks bit for bit
    ref = out.to_numpy()
    for lo in (0, (nchunks - 1) * chunk):
        hi = min(n, lo + chunk)
        assert np.array_equal(ho[lo:hi].view(np.uint64), ref[lo:hi].view(np.uint64)), "pipelined result differs"
    return {"value": n / dt / 1e9, "unit": "Gcells/s", "seconds": dt, "host_GBps": 11 * n / dt / 1e9,
            "what": f"ec_host_expr: page-locked host buffers, {nchunks} chunks of 2^25 cells, H2D / divide / D2H on three streams, double-buffered"}


def recorded_traffic(key: str, cells_per_launch: int):
    """(HBM bytes per step, where the figure comes from) of the workload's kernels.  PMC counters cannot be read
    from inside this process, so this is a RECORDED figure, not a live one: the committed rocprofv3 --pmc passes
    over this same command (profiles/traffic.json: kernel signatures, commit and method; corrected as
    MI355X_MICROARCH.md §HBM prescribes; tools/pmc_summary.py).  (None, None) when no record matches the workload
    and launch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f)
        e = rec.get(key, {})
        if e.get("cells_per_launch") == cells_per_launch:
            return e.get("hbm_bytes_per_launch"), f"recorded profiles/traffic.json[{key}] @{e.get('commit', '?')} (round {e.get('round', '?')}), not measured in this run"
    except Exception:
        pass
    return None, None


def verify_slices(env: dict, cells: int = 1 << 20):
    """Bit-exact check of the timed output against numpy's IEEE f64 divide on slices at both ends of the shard."""
    import numpy as np
    a, b, out, n = env["a"], env["b"], env["out"], env["n"]
    k = min(cells, n)
    ok = True
    for lo in sorted({0, n - k}, reverse=True):
        ha, hb, ho = a.shard(lo, k).to_numpy(), b.shard(lo, k).to_numpy(), out.shard(lo, k).to_numpy()
        with np.errstate(all="ignore"):
            exp = ha.astype(np.float64) / hb.astype(np.float64)
        ok = ok and np.array_equal(exp.view(np.uint64), ho.view(np.uint64))
    env["_head_slice"] = (ha, hb, ho)  # the shard's first cells as they left the timed region (for the oracle's check)
    return ok


class _StdoutToStderr:
    """RCCL prints a version banner on stdout when its communicator comes up; this program's stdout
    carries exactly one JSON line, so fd 1 is pointed at fd 2 while the process group initialises."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as fresh child
    processes (`python -m torch.distributed.run ... bench.py <same flags>`, one rank per GPU) and return their
    exit code.  This process only counts the devices (hipGetDeviceCount under torch.cuda.device_count()) and launches
    no GPU work of its own; it never replaces itself — the ranks are fresh child processes with their own HIP runtime,
    rank 0's JSON line reaches this process's stdout through the inherited descriptor."""
    import socket
    import subprocess

    import torch

    ndev = torch.cuda.device_count()
    if not args.single_device and ndev < args.gpus:
        sys.stderr.write(f"bench.py --gpus {args.gpus}: this box shows {ndev} HIP device(s); one rank per GPU needs "
                         f"{args.gpus} (there is no CPU fallback; --single-device --backend gloo is the 1-GPU rehearsal)\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    # the host driver of this pool only supports dmabuf IPC; RCCL fails without it (already exported on the boxes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world  # under a launcher the launcher's world size is authoritative
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    ndev = max(1, torch.cuda.device_count())
    # One rank per GPU, and only that: either this process sees every GPU of the node (dev = local_rank), or a launcher
    # masks the devices per rank and each rank sees exactly its own as device 0, or --single-device says out loud that
    # all ranks share device 0 (a rehearsal).  Anything else (8 ranks started on a 4-GPU box ...) would put several ranks
    # on one GPU and still print an N-GPU line, so it is refused.
    if args.single_device:
        dev = 0
    elif local_rank < ndev:
        dev = local_rank
    elif ndev == 1:
        dev = 0  # per-rank device masking; rank 0 checks below that the ranks' devices are distinct
    else:
        sys.exit(f"bench.py: rank {rank} (local rank {local_rank}) has no GPU of its own: {ndev} HIP devices visible for "
                 f"{world} ranks — start one rank per GPU (or --single-device for a rehearsal)")
    torch.cuda.set_device(dev)
    props = torch.cuda.get_device_properties(dev)
    dev_id = f"{props.name} uuid={getattr(props, 'uuid', '?')} pci={getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'pci_device_id', '?')}"
    use_dist = world > 1 or os.environ.get("EC_BENCH_FORCE_DIST") == "1"  # the latter: 1-rank rehearsal of the RCCL path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with _StdoutToStderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group("gloo")
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)  # brings the communicator up (and its banner out) before anything is timed
            torch.cuda.synchronize()

    import erased_cells_hip as ec
    from erased_cells_hip import sharded

    ec.init(dev)
    L = ec.lib()
    if args.variant is not None:
        ec._ffi.check(L.ec_tune_set(b"binop_variant", args.variant))
    for kv in args.tune:
        key, _, val = kv.partition("=")
        ec._ffi.check(L.ec_tune_set(key.encode(), int(val)))
    stream = torch.cuda.current_stream().cuda_stream
    ec.set_stream(stream)

    side = args.side
    rows_total = (args.rows or side) * (world if args.scaling == "weak" else 1)
    off, n = sharded.shard_range(rows_total, side, rank, world)
    total_cells = rows_total * side

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs, generated on the device, resident in HBM before timing.
    # Every workload is described by `build(k)`: allocate and fill operand set k (its own operands, temporaries and
    # outputs; set 0 carries SURVEY §8d's seeds) and return the step over it.  The timed steps ROTATE over the sets
    # (below), so no byte a step reads or writes is still in the 256 MiB Infinity Cache from the step that touched it last.
    chk = ec._ffi.check
    S = {"stream": stream}  # the launch stream; --graph points it at the capture stream while it records
    keep = []               # every set's buffers stay alive for the whole run
    scopes = {}             # div_u8_u16: set k -> (a, b, out) for the bit-exact check of the timed output
    seed_of = lambda base, k: base + 16 * k  # noqa: E731

    def new(ct, seed=None, lo=0.0, hi=0.0, cells=None):
        bf = ec.CellBuffer.empty(cells or n, ct)
        if seed is not None:
            chk(L.ec_synth_fill(ct, bf.mem.ptr, cells or n, seed, off, lo, hi, S["stream"]))
        keep.append(bf)
        return bf

    def new_mask(seed=None):
        mk = ec.Mask.empty(n)
        if seed is not None:
            chk(L.ec_synth_mask(mk.mem.ptr, n, seed, off, 30, S["stream"]))
        keep.append(mk)
        return mk

    names = ["u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64", "f32", "f64"]
    if args.workload == "div_u8_u16":
        bytes_per_cell, kernel = 11, "k_binop_direct<u8,u16,Div>" if (args.variant or 0) == 0 else "k_binop_lds<u8,u16,Div>"
        set_bytes = 11 * n
        traffic_key = "binop_div_u8_u16" if (args.variant or 0) == 0 else "binop_div_u8_u16_lds"
        wl = f"{args.rows or side}x{side} u8/u16->f64 CellBuffer divide (BASELINE configs[1]" + (")" if not args.rows else f"; the row-block of 1/{max(1, side // args.rows)} shard)")

        def build(k):
            a = new(ec.UInt8, seed_of(0x5EED0001, k), 0.0, 255.0)
            b = new(ec.UInt16, seed_of(0x5EED0002, k), 1.0, 65535.0)
            out = new(ec.Float64)
            scopes[k] = dict(a=a, b=b, out=out, n=n)
            return lambda: chk(L.ec_binop(ec.DIV, ec.UInt8, a.mem.ptr, ec.UInt16, b.mem.ptr, n, out.mem.ptr, S["stream"]))
    elif args.workload == "binop":  # any pair of cell types through the same entry point (profiles/)
        lt, rt, op = names.index(args.lt), names.index(args.rt), ["add", "sub", "mul", "div"].index(args.op)
        bytes_per_cell = ec.size_of(lt) + ec.size_of(rt) + 8
        set_bytes = bytes_per_cell * n
        kernel = f"k_binop_direct<{args.lt},{args.rt},{args.op}>"
        traffic_key = f"binop_{args.op}_{args.lt}_{args.rt}"
        wl = f"{args.rows or side}x{side} {args.lt} {args.op} {args.rt} -> f64 CellBuffer operator"

        def build(k):
            a, b = new(lt, seed_of(0x5EED0021, k), 1.0, 100.0), new(rt, seed_of(0x5EED0022, k), 1.0, 100.0)
            out = new(ec.Float64)
            return lambda: chk(L.ec_binop(op, lt, a.mem.ptr, rt, b.mem.ptr, n, out.mem.ptr, S["stream"]))
    elif args.workload == "masked_chain":
        fused = args.fused
        bytes_per_cell = 24 if fused else 42
        set_bytes = (12 + 3 + 9 + (0 if fused else 9)) * n
        kernel = "k_fused_any<4,4,4,0> (a+b)*c f32 + 3 masks, one pass" if fused else "k_masked_binop<f32,f32,Add> + k_masked_binop<f64,f32,Mul>"
        traffic_key = "masked_chain_fused" if fused else "masked_chain"
        wl = f"{side}x{side} MaskedCellBuffer f32 (a+b)*c, 30% nodata (BASELINE configs[2], " + ("fused)" if fused else "eager)")

        def build(k):
            bufs = [new(ec.Float32, seed_of(0x5EED0003 + i, k), -1000.0, 1000.0) for i in range(3)]
            masks = [new_mask(seed_of(0x5EED0013 + i, k)) for i in range(3)]
            out, m2 = new(ec.Float64), new_mask()
            if fused:
                dt4 = (C.c_uint8 * 4)(ec.Float32, ec.Float32, ec.Float32, 0)
                p4 = (C.c_void_p * 4)(bufs[0].mem.ptr, bufs[1].mem.ptr, bufs[2].mem.ptr, None)
                m4 = (C.c_void_p * 4)(masks[0].mem.ptr, masks[1].mem.ptr, masks[2].mem.ptr, None)
                keep.extend([dt4, p4, m4])
                return lambda: chk(L.ec_masked_fused(ec.ADD, ec.MUL, -1, dt4, p4, m4, None, n, out.mem.ptr, m2.mem.ptr, S["stream"]))
            t1, m1 = new(ec.Float64), new_mask()

            def step():
                chk(L.ec_masked_binop(ec.ADD, ec.Float32, bufs[0].mem.ptr, masks[0].mem.ptr, ec.Float32, bufs[1].mem.ptr,
                                      masks[1].mem.ptr, n, t1.mem.ptr, m1.mem.ptr, S["stream"]))
                chk(L.ec_masked_binop(ec.MUL, ec.Float64, t1.mem.ptr, m1.mem.ptr, ec.Float32, bufs[2].mem.ptr,
                                      masks[2].mem.ptr, n, out.mem.ptr, m2.mem.ptr, S["stream"]))
            return step
    elif args.workload == "ndvi":
        red_t = ec.Float32 if args.mixed else ec.UInt16
        traffic_key = "ndvi" + ("_fused" if args.fused else "") + ("_mixed" if args.mixed else "")
        if args.fused:
            bytes_per_cell, kernel = (14 if args.mixed else 12), ("k_fused_any<2,4,0,0>" if args.mixed else "k_fused_any<2,2,0,0>") + " (nir-red)/(nir+red), one pass"
            set_bytes = bytes_per_cell * n
        else:
            bytes_per_cell, kernel = (52 if args.mixed else 48), "k_binop_direct Sub + Add + Div (f64,f64): eager, 3 passes"
            set_bytes = (2 + ec.size_of(red_t) + 24) * n
        wl = f"{side}x{side} u16 NDVI (nir-red)/(nir+red) (BASELINE configs[4] arithmetic at raster scale), " + ("fused" if args.fused else "eager")

        def build(k):
            nir, red = new(ec.UInt16, seed_of(0x5EED0007, k), 5000.0, 40000.0), new(red_t, seed_of(0x5EED0008, k), 5000.0, 30000.0)
            out = new(ec.Float64)
            if args.fused:
                dt4 = (C.c_uint8 * 4)(ec.UInt16, red_t, ec.UInt16, red_t)
                p4 = (C.c_void_p * 4)(nir.mem.ptr, red.mem.ptr, nir.mem.ptr, red.mem.ptr)
                keep.extend([dt4, p4])
                return lambda: chk(L.ec_fused(ec.SUB, ec.DIV, ec.ADD, dt4, p4, None, n, out.mem.ptr, S["stream"]))
            t1, t2 = new(ec.Float64), new(ec.Float64)

            def step():
                chk(L.ec_binop(ec.SUB, ec.UInt16, nir.mem.ptr, red_t, red.mem.ptr, n, t1.mem.ptr, S["stream"]))
                chk(L.ec_binop(ec.ADD, ec.UInt16, nir.mem.ptr, red_t, red.mem.ptr, n, t2.mem.ptr, S["stream"]))
                chk(L.ec_binop(ec.DIV, ec.Float64, t1.mem.ptr, ec.Float64, t2.mem.ptr, n, out.mem.ptr, S["stream"]))
            return step
    elif args.workload == "evi":  # a tree deeper than two levels: 2.5*(nir-red) / (nir + 6*red - 7.5*blue + 1), 8 operators
        E = ec._ffi
        traffic_key = "evi" + ("_fused" if args.fused else "")
        sc = (E.EcValue * 4)(*[ec.CellValue.new(x).to_ec() for x in (2.5, 6.0, 7.5, 1.0)])
        U16, F64 = ec.UInt16, ec.Float64
        if args.fused:
            # default: the built-in kernel (EVI over bands of one width is in the ahead-of-time catalogue); --compiled: on the calling
            # thread before the first launch (in a pipeline the library does it in the background)
            chk(L.ec_tune_set(b"expr_fixed", 0 if (args.interpret or args.compiled) else 1))
            chk(L.ec_tune_set(b"expr_jit", 2 if args.compiled else 0))
            form = "interpreted" if args.interpret else "compiled" if args.compiled else "built-in"
            traffic_key += {"interpreted": "", "compiled": "_compiled", "built-in": "_builtin"}[form]
            bytes_per_cell, set_bytes = 14, 14 * n
            kernel = {"interpreted": "k_expr<2,2,2,0> (interpreter)", "compiled": "ec_expr_jit (the program compiled with hiprtc)",
                      "built-in": "k_expr_fixed<EVI, 2> (ahead-of-time catalogue)"}[form] + ": 8 operators over 3 u16 bands, one pass"
            Sx, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
            prog = [(ec.SUB, Sx(0), Sx(1), 0), (ec.MUL, R(0), K(0), 0),     # r0 = (nir - red) * 2.5
                    (ec.MUL, Sx(1), K(1), 1), (ec.ADD, Sx(0), R(1), 1),     # r1 = nir + red * 6
                    (ec.MUL, Sx(2), K(2), 2), (ec.SUB, R(1), R(2), 1),      # r1 = r1 - blue * 7.5
                    (ec.ADD, R(1), K(3), 1), (ec.DIV, R(0), R(1), 0)]       # r0 = r0 / (r1 + 1)
            st = (E.EcExprStep * len(prog))(*[E.EcExprStep(*q) for q in prog])
            dt3 = (C.c_uint8 * 3)(U16, U16, U16)
        else:
            # the reference's eager evaluation: every operator one pass over f64 temporaries
            bytes_per_cell = (2 + 2 + 8) + (8 + 8) + (2 + 8) + (2 + 8 + 8) + (2 + 8) + (8 + 8 + 8) + (8 + 8) + (8 + 8 + 8)  # 130
            set_bytes = (6 + 8 * 8) * n
            kernel = "k_binop_direct / k_binop_scalar x 8 (f64 temporaries): eager, 8 passes"
        wl = f"{side}x{side} u16 EVI 2.5(nir-red)/(nir+6red-7.5blue+1), 8 operators, " + (("one pass (ec_expr, " + form + ")") if args.fused else "eager")

        def build(k):
            nir, red, blue = (new(U16, seed_of(0x5EED0031 + i, k), 2000.0 + 3000.0 * (2 - i), 20000.0 + 10000.0 * (2 - i)) for i in range(3))
            out = new(F64)
            if args.fused:
                p3 = (C.c_void_p * 3)(nir.mem.ptr, red.mem.ptr, blue.mem.ptr)
                keep.append(p3)
                return lambda: chk(L.ec_expr(dt3, p3, 3, sc, 4, st, len(prog), n, out.mem.ptr, S["stream"]))
            t = [new(F64) for _ in range(7)]  # one temporary per operator, as the reference allocates

            def step():
                s = S["stream"]
                chk(L.ec_binop(ec.SUB, U16, nir.mem.ptr, U16, red.mem.ptr, n, t[0].mem.ptr, s))
                chk(L.ec_binop_scalar(ec.MUL, F64, t[0].mem.ptr, n, C.byref(sc[0]), t[1].mem.ptr, s))
                chk(L.ec_binop_scalar(ec.MUL, U16, red.mem.ptr, n, C.byref(sc[1]), t[2].mem.ptr, s))
                chk(L.ec_binop(ec.ADD, U16, nir.mem.ptr, F64, t[2].mem.ptr, n, t[3].mem.ptr, s))
                chk(L.ec_binop_scalar(ec.MUL, U16, blue.mem.ptr, n, C.byref(sc[2]), t[4].mem.ptr, s))
                chk(L.ec_binop(ec.SUB, F64, t[3].mem.ptr, F64, t[4].mem.ptr, n, t[5].mem.ptr, s))
                chk(L.ec_binop_scalar(ec.ADD, F64, t[5].mem.ptr, n, C.byref(sc[3]), t[6].mem.ptr, s))
                chk(L.ec_binop(ec.DIV, F64, t[1].mem.ptr, F64, t[6].mem.ptr, n, out.mem.ptr, s))
            return step
    else:
        bytes_per_cell, kernel, set_bytes = 2, "k_min_max_partials<u16>", 2 * n
        traffic_key = "minmax"
        wl = f"{side}x{side} u16 min_max, row-sharded, all-reduce of 2 int64 keys (BASELINE configs[3] shape)"
        if side == 65536:
            wl += (f"; configs[3]'s whole 8.6 GB raster, {rows_total // world} rows per rank" +
                   (" - on ONE GPU here; at 8 GPUs a rank's shard is 8192 x 65536 cells, 1.07 GB" if world == 1 else ""))
        keys = torch.empty(2, dtype=torch.int64, device="cuda")

        def build(k):
            a = new(ec.UInt16, seed_of(0x5EED0006, k), 1.0, 65534.0)

            def step():
                chk(L.ec_min_max_keys(ec.UInt16, a.mem.ptr, None, n, keys.data_ptr(), S["stream"]))
                if use_dist:
                    dist.all_reduce(keys, op=dist.ReduceOp.MAX)
            return step

    # How many operand sets.  Between two uses of a set, (sets - 1) x set_bytes of other sets' bytes pass through the memory
    # system; with that at >= 1 GiB (four times the Infinity Cache) nothing of a set is on-die when its turn comes again.  A
    # set that is itself >= 1 GiB (the whole 16384² divide: 2.95 GB) is rotated over 4 sets all the same, for its 256 MiB
    # u8 operand, which the library would otherwise find in the cache (cache_plan, csrc/ec_runtime.hpp).  At N = 8 a rank's
    # set is 369 MB (operands 32 + 64 MiB, output 256 MiB — all of which could sit on-die): 4 sets, 1.1 GB between two uses.
    GIB = 1 << 30
    nsets = args.sets if args.sets else max(4, min(96, 1 + -(-GIB // max(1, set_bytes))))
    steps_of = [build(k) for k in range(nsets)]
    torch.cuda.synchronize()
    turn = {"i": 0}

    def step():  # the next set's turn: round-robin through ramp, warm-up and timed steps alike
        steps_of[turn["i"] % nsets]()
        turn["i"] += 1

    # ---- clock ramp + warm-up, then EXACTLY `steps` timed steps between barrier+synchronize.
    # The first ≈25 ms of launches after idle run ≈5 % slow while the GPU's clocks ramp
    # (profiles/r01/warmup_sensitivity.txt).  Untimed ramp launches run first until both --ramp launches
    # and --ramp-ms of wall time have passed (a 1/8 shard's step is only ≈60 µs); their number is
    # disclosed as config.clock_ramp_steps (and config.untimed_steps_before_timing = ramp + W).  Then the W warm-up
    # steps the caller asked for.
    # The collector is emptied BEFORE the ramp (a collection between warm-up and timing would idle the GPU for tens of
    # milliseconds and undo the ramp) and stays off until the timed region has ended: a collector pause inside a 1 ms
    # timed region (K steps of a 1/8 shard) would be a tenth of it.
    import gc
    gc.collect()
    gc.disable()
    if use_dist:
        barrier()  # ranks enter the ramp together, so they reach the timing barrier within microseconds of each other:
                   # a rank that waited there for long would start its timed steps on an idle, down-clocked GPU
    ramp = 0
    t_ramp = time.perf_counter()
    while args.ramp > 0 and (ramp < args.ramp or (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms):
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ramp += 20
    for _ in range(args.warmup):
        step()
    graph = None
    if args.graph and args.workload == "minmax" and use_dist:
        sys.exit("--graph does not capture the all-reduce of the sharded minmax workload")
    first_timed = turn["i"]
    if args.graph:
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        ec.set_stream(cap.cuda_stream)
        S["stream"] = cap.cuda_stream  # the steps read the stream at call time
        chk(L.ec_prepare_stream(S["stream"]))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap):
            for _ in range(args.steps):
                step()
        S["stream"] = torch.cuda.current_stream().cuda_stream
        ec.set_stream(S["stream"])
        graph.replay()  # first replay uploads the graph
        torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    torch.cuda.synchronize()              # this rank's K steps are complete on the device
    elapsed = time.perf_counter() - t0    # per-rank time of exactly K steps; the MAX over ranks is reported
    gc.enable()
    barrier()                             # closing bracket: every rank is done before anything else happens
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream
    last_set = (first_timed + args.steps - 1) % nsets  # the set the last timed step ran over

    # ---- the timed output is checked, outside `value`: a slice at each end of this rank's shard against
    # numpy's IEEE f64 arithmetic on the same operands (the oracle itself checks it in the cpu_baseline leg).  Every set
    # the timed region wrote is checked (at most nsets of them), the last one last: its head slice goes to the oracle.
    scope = None
    verified = None
    if args.workload == "div_u8_u16":
        touched = sorted({(first_timed + i) % nsets for i in range(min(args.steps, nsets))}, key=lambda k: k == last_set)
        verified = True
        for k in touched:
            scope = scopes[k]
            verified = verify_slices(scope) and verified

    # ---- beside the headline, on EVERY rank and outside `value`: the same step over ONE set again and again — the loop
    # the contract's wording suggests (one batch, K steps), what rounds 1-3 reported as the headline, and what a caller
    # who iterates over a resident raster sees.  Operands that fit the Infinity Cache are then served on-die from the
    # second step on (the 256 MiB u8 operand at N = 1; both operands of a 1/8 shard), so this figure divides bytes that
    # never crossed HBM by the HBM peak: it is reported as `roofline.cache_resident_loop`, never as `frac`.
    resident_ms, build_up = float("nan"), []
    if not args.no_resident_loop and not args.graph:
        # The loop's residency builds up slowly: from a cold start (the rotating steps leave nothing of this set on-die) the one-set
        # loop needs some 40 launches before the operands that fit the cache are served from it at the steady rate, whatever is done
        # to prime them (three read-only passes over the operands beforehand changed nothing: profiles/r04/resident_loop_build_up.md).
        # 40 untimed launches, then 20 timed; the launch times of the untimed ones are kept in `build_up_ms` (groups of 10).
        reps, build_up = 20, []
        for _ in range(4):
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(10):
                steps_of[last_set]()
            g1.record()
            build_up.append((g0, g1))
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(reps):
            steps_of[last_set]()
        r1.record()
        torch.cuda.synchronize()
        resident_ms = r0.elapsed_time(r1) / reps
        build_up = [a.elapsed_time(b) / 10 for a, b in build_up]
        barrier()

    # per-rank record -> every rank: [elapsed s, device ms, cells, verified, resident-loop ms per step]
    mine = torch.tensor([elapsed, dev_ms, float(n), 1.0 if verified in (None, True) else 0.0, resident_ms], dtype=torch.float64)
    if use_dist:
        dev_t = "cuda" if args.backend == "nccl" else "cpu"
        gathered = [torch.empty(5, dtype=torch.float64, device=dev_t) for _ in range(world)]
        dist.all_gather(gathered, mine.to(dev_t))
        per_rank = [g.cpu().tolist() for g in gathered]
    else:
        per_rank = [mine.tolist()]
    if use_dist:
        dev_ids = [None] * world
        dist.all_gather_object(dev_ids, dev_id)
    else:
        dev_ids = [dev_id]
    identifiable = all("uuid=?" not in d or "pci=?" not in d for d in dev_ids)  # a build without uuid / PCI ids cannot tell
    if rank == 0 and world > 1 and not args.single_device and identifiable and len(set(dev_ids)) != world:
        sys.exit(f"bench.py: the {world} ranks do not sit on {world} distinct GPUs: {dev_ids}")
    elapsed = max(r[0] for r in per_rank)          # MAX over ranks
    slowest = max(range(len(per_rank)), key=lambda i: per_rank[i][1])
    dev_ms = per_rank[slowest][1]
    if verified is not None:
        verified = all(r[3] == 1.0 for r in per_rank)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        launch_ms = dev_ms / args.steps
        n_slowest = int(per_rank[slowest][2])
        gbps = lambda cells, ms: bytes_per_cell * cells / (ms * 1e-3) / 1e9  # noqa: E731
        achieved = gbps(n_slowest, launch_ms)  # the slowest GPU's launch: its bytes / its time
        traffic, traffic_source = recorded_traffic(traffic_key, n_slowest) if world == 1 else (None, None)
        per_gpu = []
        for i, r in enumerate(per_rank):
            g = {"rank": i, "device": dev_ids[i], "cells": int(r[2]), "launch_ms": r[1] / args.steps,
                 "frac": gbps(r[2], r[1] / args.steps) / HBM_PEAK_GBPS}
            if r[4] == r[4]:  # not NaN: the one-set loop ran
                g["cache_resident_loop"] = {"launch_ms": r[4], "frac": gbps(r[2], r[4]) / HBM_PEAK_GBPS}
            per_gpu.append(g)
        res = {
            "metric": METRIC if args.workload == "div_u8_u16" else f"Gcells/s ({args.workload})",
            "value": total_cells / (elapsed / args.steps) / 1e9,
            "unit": "Gcells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64" if args.workload != "minmax" else "u16",
            "data": "synthetic",
            "config": {"workload": wl, "rows": rows_total, "cols": side, "cells": total_cells,
                       "cells_per_gpu": n, "cells_per_rank": [int(r[2]) for r in per_rank], "sharding": "contiguous row-block per rank, no data-path collective",
                       "inputs": "splitmix64 counter streams generated on device, resident in HBM",
                       "operand_sets": nsets, "operand_set_bytes_per_rank": int(set_bytes),
                       "rotation": f"step i runs over operand set i mod {nsets} (own operands and own output, every rank its own sets): "
                                   f"{(nsets - 1) * set_bytes / GIB:.2f} GiB of other sets' bytes pass between two uses of a byte, so every "
                                   "algorithmic byte of a timed step moves from / to HBM, none from the 256 MiB Infinity Cache",
                       "kernel_variant": "lds" if (args.variant or 0) == 1 else "direct", "tune": args.tune,
                       "clock_ramp_steps": ramp, "untimed_steps_before_timing": ramp + args.warmup,
                       "launch": "hipGraph of the K steps" if args.graph else "K stream launches"},
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_cell": bytes_per_cell, "cells_per_launch": n_slowest,
                         "launch_ms": launch_ms, "timer": "hipEvent pair on the launch stream over the timed region / steps"
                                                          + ("; the slowest rank's launch" if world > 1 else ""),
                         "all_hbm": bool((nsets - 1) * set_bytes >= GIB),
                         "per_gpu": per_gpu},
        }
        res_ms = [r[4] for r in per_rank if r[4] == r[4]]
        if len(res_ms) == len(per_rank):
            i_slow = max(range(len(per_rank)), key=lambda i: per_rank[i][4])
            r_ms, r_cells = per_rank[i_slow][4], per_rank[i_slow][2]
            res["roofline"]["cache_resident_loop"] = {
                "launch_ms": r_ms, "achieved": gbps(r_cells, r_ms), "frac": gbps(r_cells, r_ms) / HBM_PEAK_GBPS,
                "Gcells_per_s": total_cells / (r_ms * 1e-3) / 1e9, "operand_sets": 1, "launches": "40 untimed + 20 timed",
                "build_up_ms_rank0": build_up,
                "what": "NOT an HBM figure: the same step over ONE operand set, launch after launch (the slowest rank's).  Operands that fit "
                        "the 256 MiB Infinity Cache are served on-die from the second launch on (the 256 MiB u8 operand of the whole "
                        "raster; both operands of a 1/8 shard), so part of these bytes never crossed HBM.  Rounds 1-3 reported this as "
                        "the headline; since round 4 `value`, `ms_per_step` and `roofline.frac` come from the rotating sets above."}
        if verified is not None:
            res["verified"] = verified
            res["config"]["verified_how"] = ("after the timed region every rank compares 2^20 cells at each end of its output shard, for every "
                                             "operand set the timed steps wrote, bit for bit with numpy's f64 divide of the same operands")
        if args.e2e and world == 1 and args.workload == "div_u8_u16":
            import numpy as np
            a, b, out = scope["a"], scope["b"], scope["out"]
            ha, hb = a.to_numpy(), b.to_numpy()
            t1 = time.perf_counter()
            r = (ec.CellBuffer.from_vec(ha) / ec.CellBuffer.from_vec(hb)).to_numpy()
            dt = time.perf_counter() - t1
            res["end_to_end_pcie"] = {"value": n / dt / 1e9, "unit": "Gcells/s", "seconds": dt,
                                      "what": "from_vec(u8)+from_vec(u16) over PCIe, divide, to_vec(f64) back; pageable host memory"}
            del r
            res["end_to_end_pcie_pipelined"] = e2e_pipelined(torch, ec, L, a, b, out, n)
        if world == 1 and not args.no_reference_streams and args.workload == "div_u8_u16":
            # SURVEY §8(d) "empirical ceiling": what plain streams of the same buffers reach on this box, so the
            # fraction of *achievable* bandwidth can be read next to the fraction of the nominal 8 TB/s.  Outside
            # the timed region; same launch shape family, no divide; rotating over the same operand sets (all-HBM, like `frac`).
            def rate(fn_of, nbytes, reps=60):
                for i in range(reps // 2):
                    fn_of(i % nsets)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    fn_of(i % nsets)
                e1.record()
                torch.cuda.synchronize()
                return nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9

            zero = ec.CellValue.new(0.0).to_ec()
            keys = torch.empty(2, dtype=torch.int64, device="cuda")
            sc_ = scopes
            res["roofline"]["reference_streams"] = {
                "same_mix_add_u8_u16_GBps": rate(lambda k: chk(L.ec_binop(ec.ADD, ec.UInt8, sc_[k]["a"].mem.ptr, ec.UInt16, sc_[k]["b"].mem.ptr, n,
                                                                          sc_[k]["out"].mem.ptr, S["stream"])), 11 * n),
                "write_only_fill_f64_GBps": rate(lambda k: chk(L.ec_fill(ec.Float64, sc_[k]["out"].mem.ptr, n, C.byref(zero), S["stream"])), 8 * n),
                "read_only_min_max_f64_GBps": rate(lambda k: chk(L.ec_min_max_keys(ec.Float64, sc_[k]["out"].mem.ptr, None, n, keys.data_ptr(),
                                                                                   S["stream"])), 8 * n),
                "what": "library kernels over the same rotating operand sets, untimed by `value`: 3 B read + 8 B written with an add "
                        "instead of the divide; 8 B/cell written; 8 B/cell read",
            }
        if world == 1 and not args.no_cpu_baseline and args.workload == "div_u8_u16":
            res["cpu_baseline"] = cpu_baseline(side, args.cpu_seconds)
            # the oracle as the checker of the timed output (same leg, outside `value`): 2^20 cells, bit for bit,
            # against both of its forms (reference-shaped and typed loop)
            import numpy as np
            from oracle import eco
            ha, hb, ho = scope["_head_slice"]  # downloaded right after the timed region, before the reference streams reuse `out`
            ok = np.array_equal(eco.f_binop(eco.DIV, ha, hb).view(np.uint64), ho.view(np.uint64)) and \
                np.array_equal(eco.binop(eco.DIV, ha[:65536], hb[:65536]).view(np.uint64), ho[:65536].view(np.uint64))
            res["cpu_baseline"]["oracle_check_of_timed_output"] = bool(ok)
            res["verified"] = bool(res.get("verified", True) and ok)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

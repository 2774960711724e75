# A/B: the built-in expression kernels over the whole tile (default) against chunk by chunk (EC_FIXED_CHUNKED=1), interleaved.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O
cd $R
B=$R/erased-cells_amd/liberased_cells_hip_fixedchunk.so
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-resident-loop --workload evi --fused > $O/evi_tile_$rep.json 2>> $O/err
  EC_HIP_LIB=$B python bench.py --no-cpu-baseline --no-resident-loop --workload evi --fused > $O/evi_chunk_$rep.json 2>> $O/err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04n/evi_*.json")):
    r = json.load(open(f)); rf = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5), rf["kernel"][:40])
PY
python - <<'PY'
# (a + b) * c and NDVI through ec_expr, both libraries, one process each
import subprocess, sys, os
code = r'''
import sys, ctypes as C
sys.path.insert(0, "erased-cells_amd/python")
import torch, erased_cells_hip as ec
torch.cuda.set_device(0); ec.init(0); L = ec.lib(); chk = ec._ffi.check; E = ec._ffi
s = torch.cuda.current_stream().cuda_stream
n = 16384 * 16384
chk(L.ec_tune_set(b"expr_jit", 0))
sets = []
for k in range(4):
    bs = [ec.CellBuffer.empty(n, ec.UInt16) for _ in range(3)]
    for i, b in enumerate(bs): chk(L.ec_synth_fill(ec.UInt16, b.mem.ptr, n, 77 + 16 * k + i, 0, 1.0, 30000.0, s))
    sets.append((bs, ec.CellBuffer.empty(n, ec.Float64)))
progs = {"(a+b)*c": ([(ec.ADD, 0, 1, 0), (ec.MUL, 4, 2, 0)], 3, 14), "ndvi": ([(ec.SUB, 0, 1, 0), (ec.ADD, 0, 1, 1), (ec.DIV, 4, 5, 0)], 2, 12)}
for name, (prog, ns, bpc) in progs.items():
    st = (E.EcExprStep * len(prog))(*[E.EcExprStep(*q) for q in prog])
    dt = (C.c_uint8 * ns)(*([ec.UInt16] * ns))
    ps = [(C.c_void_p * ns)(*[b.mem.ptr for b in bs[:ns]]) for bs, _ in sets]
    def run(k):
        for i in range(k): chk(L.ec_expr(dt, ps[i & 3], ns, None, 0, st, len(prog), n, sets[i & 3][1].mem.ptr, s))
    run(120); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(100); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print(f"{name}: {ms:.4f} ms frac {bpc * n / ms / 1e6 / 8000:.4f}")
'''
for rep in range(2):
    for lib in ("", os.environ["GRAFT_REPO_ROOT"] + "/erased-cells_amd/liberased_cells_hip_fixedchunk.so"):
        env = dict(os.environ); 
        if lib: env["EC_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("chunked" if lib else "tile   ", r.stdout.replace("\n", " | "), r.stderr[-200:] if r.returncode else "")
PY

# Round 4: third store sweep (wide loads through LDS, waves that wait for their stores), the counters of the store variants,
# what the library's size costs at load time, the fill with capped occupancy.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
./tools/tune_store3 16384 9 10 > $O/tune_store_v3.log 2> $O/tune_store_v3.err || { tail -5 $O/tune_store_v3.err; }
python - <<'PY'
import re
rows = [l.rstrip("\n") for l in open("gpurun_out/r04e/tune_store_v3.log") if l[:3] in ("wr ", "mix", "ref", "rd1")]
key = lambda l: -float(re.search(r"(\d\.\d+)\s+\S+$", l).group(1))
for kind in ("wr ", "mix"):
    print("\n".join(sorted((l for l in rows if l.startswith(kind)), key=key))); print()
PY
python tools/load_latency.py > $O/load_latency.md 2> $O/load_latency.err || tail -5 $O/load_latency.err
cat $O/load_latency.md
for kb in 64 48 32 0; do
  python - $kb <<'PY'
import sys, ctypes as C
sys.path.insert(0, "erased-cells_amd/python")
import torch, erased_cells_hip as ec
kb = int(sys.argv[1])
torch.cuda.set_device(0); ec.init(0); L = ec.lib(); chk = ec._ffi.check
chk(L.ec_tune_set(b"write_lds_kb", kb))
n = 16384 * 16384
outs = [ec.CellBuffer.empty(n, ec.Float64) for _ in range(2)]
zero = ec.CellValue.new(0.0).to_ec()
s = torch.cuda.current_stream().cuda_stream
def run(k):
    for i in range(k): chk(L.ec_fill(ec.Float64, outs[i & 1].mem.ptr, n, C.byref(zero), s))
run(100); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(100); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 100
print(f"fill Float64 16384^2, write_lds_kb = {kb}: {ms:.4f} ms, {8 * n / ms / 1e6:.0f} GB/s, {8 * n / ms / 1e6 / 8000:.4f} of peak")
PY
done
bash tools/jobs/r04store_pmc.sh > $O/store_pmc.md 2> $O/store_pmc.err || tail -5 $O/store_pmc.err
cat $O/store_pmc.md

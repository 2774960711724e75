# A/B: k_fused_any chunk by chunk (default now) against over the whole tile (…_fusedtile.so), interleaved; the synchronous-result latency
# with and without polling; parity of the fused kernels and the sync entry points.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_instantiations.py tests/test_gpu_parity.py tests/test_gpu_reference_kats.py -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
B=$R/erased-cells_amd/liberased_cells_hip_fusedtile.so
for rep in 1 2; do
  for wl in "ndvi --workload ndvi --fused" "ndvimixed --workload ndvi --fused --mixed" "chain --workload masked_chain --fused"; do
    set -- $wl; key=$1; shift
    python bench.py --no-cpu-baseline --no-resident-loop "$@" > $O/${key}_chunk_$rep.json 2>> $O/err
    EC_HIP_LIB=$B python bench.py --no-cpu-baseline --no-resident-loop "$@" > $O/${key}_tile_$rep.json 2>> $O/err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04o/*_*.json")):
    r = json.load(open(f)); rf = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5))
PY
python tools/sync_result_latency.py > $O/sync_result_latency.txt 2>> $O/err
cat $O/sync_result_latency.txt

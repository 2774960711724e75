# A/B: the short divide's tile code staged over the whole tile (default) against chunk by chunk (EC_DIV_STAGE=2), interleaved; then the suite.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O
cd $R
B=$R/erased-cells_amd/liberased_cells_hip_divstage2.so
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop > $O/bench_stage1_$rep.json 2>> $O/err
  EC_HIP_LIB=$B python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop > $O/bench_stage2_$rep.json 2>> $O/err
done
python bench.py --no-cpu-baseline --no-reference-streams --no-resident-loop --workload binop --lt u8 --rt u16 --op add > $O/bench_add.json 2>> $O/err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04m/bench_*.json")):
    r = json.load(open(f)); rf = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5), r.get("verified"))
PY
python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1

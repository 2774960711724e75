# Occupancy caps for the 8-byte binop / scalar families: the eager chains (what the reference's operators run) with the rule and without.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_instantiations.py -q -m gpu -s -k "binop or scalar or every_pair or divide" > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault|Error" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python tools/write_heavy_caps.py > $O/write_heavy_caps.md 2> $O/err || tail -3 $O/err
for rep in 1 2; do
  for wl in "evi --workload evi" "ndvi --workload ndvi" "chain --workload masked_chain"; do
    set -- $wl; key=$1; shift
    python bench.py --no-cpu-baseline --no-resident-loop "$@" > $O/${key}_rule_$rep.json 2>> $O/err
    python bench.py --no-cpu-baseline --no-resident-loop "$@" --tune binop_lds_kb=0 --tune scalar_lds_kb=0 > $O/${key}_nocap_$rep.json 2>> $O/err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04s/*_*.json")):
    r = json.load(open(f)); rf = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 2), "Gcells/s frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 4))
PY

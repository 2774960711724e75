# The NaN rule of cv_bin_op! tested once per chunk / tile instead of per cell: parity, then the kernel table and the float workloads.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04u; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_instantiations.py tests/test_gpu_fullsize.py tests/test_gpu_reference_kats.py -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault|Error" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python tools/kernel_table.py > $O/kernel_table.md 2> $O/err
grep -E "^\| (binop|masked|fused|binop_scalar)" $O/kernel_table.md | cut -c1-140
python tools/write_heavy_caps.py > $O/write_heavy_caps.md 2>> $O/err; head -8 $O/write_heavy_caps.md
for wl in "--workload binop --lt f32 --rt f32 --op add" "--workload masked_chain" "--workload masked_chain --fused" "--workload ndvi --fused --mixed" "--workload evi"; do
  python bench.py --no-cpu-baseline --no-resident-loop $wl >> $O/bench.jsonl 2>> $O/err
done
python - <<'PY'
import json
for l in open("gpurun_out/r04u/bench.jsonl"):
    r = json.loads(l); print(r["config"]["workload"][:90], "|", round(r["value"], 1), round(r["roofline"]["frac"], 4))
PY

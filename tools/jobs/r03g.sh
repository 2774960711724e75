# Expression-program kernel (k_expr): its tests first, then the whole GPU suite, then EVI eager vs one pass and the headline.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_instantiations.py -x -q -m gpu -k "expr or lazy" > $O/pytest_expr.log 2>&1 || { tail -60 $O/pytest_expr.log; exit 1; }
tail -3 $O/pytest_expr.log
python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -10 $O/pytest_gpu.log
rm -f $O/bench_evi.jsonl
for w in "--workload evi" "--workload evi --fused" "--workload evi --fused --side 8192" "--workload ndvi --fused" "--workload masked_chain --fused"; do
  python bench.py $w --no-cpu-baseline >> $O/bench_evi.jsonl 2>> $O/bench_evi.err
done
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03g/bench_n1.json")); rf = r["roofline"]
print("headline", round(r["value"], 1), round(rf["frac"], 4), rf.get("fresh_inputs", {}).get("frac"))
for l in open("gpurun_out/r03g/bench_evi.jsonl"):
    r = json.loads(l)
    print(f"{r['config']['workload'][:78]:78s} {r['value']:8.1f} Gcells/s  frac {r['roofline']['frac']:.3f} ms {r['ms_per_step']:.3f}")
PY

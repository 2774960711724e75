# Round-3: GPU suite + bench lines on the load-policy build (policy_arms / cache_plan, k_fused_any only).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -14 $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cat $O/bench_n1.json
python bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2>> $O/bench_n1.err
for w in "--rows 2048" "--workload masked_chain" "--workload masked_chain --fused" "--workload minmax" "--workload minmax --side 32768" "--workload ndvi" "--workload ndvi --fused" "--workload ndvi --fused --mixed" "--workload binop --lt u16 --rt u16 --op add" "--workload binop --lt f32 --rt f32 --op add" "--workload binop --lt u8 --rt u8 --op add"; do
  python bench.py $w --no-cpu-baseline >> $O/bench_all_workloads.jsonl 2>> $O/bench_all.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r03d/bench_all_workloads.jsonl"):
    r = json.loads(l)
    print(f"{r['config']['workload'][:70]:70s} {r['value']:8.1f} Gcells/s  frac {r['roofline']['frac']:.3f}")
PY

# Round 4: the whole GPU suite on the library as it ships (value stores sc1 nt, mask stores nt, pure-write launches capped at two
# workgroups per CU, built-in expression kernels), the headline under rocprofv3, the cache-policy A/B, load latency.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -30; }
tail -2 $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2>> $O/err
python bench.py --no-cpu-baseline --rows 2048 > $O/bench_shard8.json 2>> $O/err
python tools/load_latency.py > $O/load_latency.md 2> $O/load_latency.err || tail -5 $O/load_latency.err
cat $O/load_latency.md
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_default --output-format csv -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_under_rocprof.json 2> $O/prof_default.err || tail -5 $O/prof_default.err
rocprofv3 --kernel-trace --stats -d $O/prof_k200 --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench_k200_under_rocprof.json 2> $O/prof_k200.err || tail -5 $O/prof_k200.err
rocprofv3 --kernel-trace --stats -d $O/prof_rot --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-resident-loop --no-reference-streams > $O/bench_rotating_only_under_rocprof.json 2> $O/prof_rot.err || tail -5 $O/prof_rot.err
cd $R
for d in prof_default prof_k200 prof_rot; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); cp "$f" $O/${d}_kernel_stats.csv; find $O/$d -name '*kernel_trace.csv' -delete
  echo "== $d"; head -6 $O/${d}_kernel_stats.csv | cut -c1-260
done
python - <<'PY'
import json
for f in ("bench_n1", "bench_n1_driver_flags", "bench_shard8", "bench_default_under_rocprof", "bench_k200_under_rocprof", "bench_rotating_only_under_rocprof"):
    r = json.load(open(f"gpurun_out/r04f/{f}.json")); rf = r["roofline"]
    print(f, round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5),
          "resident", round(rf.get("cache_resident_loop", {}).get("frac", 0), 4), r.get("verified"),
          {k: round(v) for k, v in rf.get("reference_streams", {}).items() if k != "what"}, r.get("cpu_baseline", {}).get("value"))
PY
bash tools/jobs/r04c.sh > $O/cache_plan_ab.md 2> $O/cache_plan_ab.err || tail -5 $O/cache_plan_ab.err
cat $O/cache_plan_ab.md

# Round 4: the headline with the primed one-set loop, under rocprofv3; the cache-conflict probe; HBM traffic of every workload.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
python bench.py > $O/bench_n1.json 2> $O/err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2>> $O/err
python bench.py --no-cpu-baseline --rows 2048 > $O/bench_shard8.json 2>> $O/err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_default --output-format csv -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_under_rocprof.json 2> $O/prof_default.err || tail -5 $O/prof_default.err
rocprofv3 --kernel-trace --stats -d $O/prof_rot --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-resident-loop --no-reference-streams > $O/bench_rotating_only_under_rocprof.json 2> $O/prof_rot.err || tail -5 $O/prof_rot.err
cd $R
for d in prof_default prof_rot; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); cp "$f" $O/${d}_kernel_stats.csv; find $O/$d -name '*kernel_trace.csv' -delete
  echo "== $d"; head -4 $O/${d}_kernel_stats.csv | cut -c1-230
done
python - <<'PY'
import json
for f in ("bench_n1", "bench_n1_driver_flags", "bench_shard8", "bench_default_under_rocprof", "bench_rotating_only_under_rocprof"):
    r = json.load(open(f"gpurun_out/r04j/{f}.json")); rf = r["roofline"]
    print(f, round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5),
          "resident", round(rf.get("cache_resident_loop", {}).get("frac", 0), 4), r.get("verified"),
          {k: round(v) for k, v in rf.get("reference_streams", {}).items() if k != "what"}, r.get("cpu_baseline", {}).get("value"))
PY
python tools/cache_conflict_probe.py > $O/cache_conflict_probe.md 2>> $O/err || tail -3 $O/err
cat $O/cache_conflict_probe.md
EC_COMMIT=$(cat $R/.ec_commit 2>/dev/null || echo r04) bash tools/jobs/r04pmc.sh > $O/pmc.log 2> $O/pmc.err || tail -5 $O/pmc.err
tail -25 $O/pmc.log

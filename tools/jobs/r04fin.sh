# Round 4, the state that ships: the whole GPU suite (uncaptured), the headline three ways, the kernel table, every bench workload,
# rocprofv3 kernel traces of the default command and of the rotating steps alone.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04fin; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/bench_n1.json 2> $O/err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2>> $O/err
python bench.py --no-cpu-baseline --rows 2048 > $O/bench_shard8.json 2>> $O/err
python bench.py --no-cpu-baseline --rows 2048 --graph > $O/bench_shard8_graph.json 2>> $O/err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_default --output-format csv -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_under_rocprof.json 2> $O/prof_default.err || tail -5 $O/prof_default.err
rocprofv3 --kernel-trace --stats -d $O/prof_rot --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-resident-loop --no-reference-streams > $O/bench_rotating_only_under_rocprof.json 2> $O/prof_rot.err || tail -5 $O/prof_rot.err
cd $R
for d in prof_default prof_rot; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); cp "$f" $O/${d}_kernel_stats.csv; find $O/$d -name '*kernel_trace.csv' -delete
  echo "== $d"; head -3 $O/${d}_kernel_stats.csv | cut -c1-230
done
python tools/kernel_table.py > $O/kernel_table.md 2>> $O/err
: > $O/bench_all_workloads.jsonl
for wl in "--workload masked_chain" "--workload masked_chain --fused" "--workload minmax" "--workload minmax --side 32768" "--workload minmax --side 65536" \
          "--workload ndvi" "--workload ndvi --fused" "--workload ndvi --fused --mixed" "--workload evi" "--workload evi --fused" "--workload evi --fused --interpret" \
          "--workload evi --fused --compiled" "--workload binop --lt u8 --rt u16 --op add" "--workload binop --lt u16 --rt u16 --op add" \
          "--workload binop --lt f32 --rt f32 --op add" "--workload binop --lt u8 --rt u8 --op div" "--workload binop --lt u8 --rt u8 --op add" "--side 32768" "--side 8192" "--side 4096"; do
  python bench.py --no-cpu-baseline $wl >> $O/bench_all_workloads.jsonl 2>> $O/err
done
python - <<'PY'
import json
for f in ("bench_n1", "bench_n1_driver_flags", "bench_shard8", "bench_shard8_graph", "bench_default_under_rocprof", "bench_rotating_only_under_rocprof"):
    r = json.load(open(f"gpurun_out/r04fin/{f}.json")); rf = r["roofline"]
    print(f, round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5),
          "resident", round(rf.get("cache_resident_loop", {}).get("frac", 0), 4), [round(x, 4) for x in rf.get("cache_resident_loop", {}).get("build_up_ms_rank0", [])], r.get("verified"),
          {k: round(v) for k, v in rf.get("reference_streams", {}).items() if k != "what"}, r.get("cpu_baseline", {}).get("value"))
for l in open("gpurun_out/r04fin/bench_all_workloads.jsonl"):
    r = json.loads(l); rf = r["roofline"]
    print(r["config"]["workload"][:100], "|", round(r["value"], 1), "Gcells/s frac", round(rf["frac"], 4), "loop", round(rf.get("cache_resident_loop", {}).get("frac", 0), 4), "sets", r["config"]["operand_sets"])
PY
grep -E "^\| (binop (Add|Div) UInt8|fill|min_max UInt8 \||mask_counts|first_diff|expr)" $O/kernel_table.md | cut -c1-160

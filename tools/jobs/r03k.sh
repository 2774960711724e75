# Round-3 evidence on the FINAL build (expression programs interpreted and compiled): GPU suite, bench lines, the EVI cost table,
# kernel-instantiation coverage, the two-column kernel table, rocprof kernel stats of bench.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -10 $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2>> $O/bench_n1.err
rm -f $O/bench_all_workloads.jsonl
for w in "--rows 2048" "--rows 2048 --graph" "--workload masked_chain" "--workload masked_chain --fused" "--workload minmax" "--workload minmax --side 32768" "--workload minmax --side 65536 --steps 20 --warmup 3" "--workload ndvi" "--workload ndvi --fused" "--workload ndvi --mixed" "--workload ndvi --fused --mixed" "--workload evi" "--workload evi --fused --interpret" "--workload evi --fused" "--workload binop --lt u16 --rt u16 --op add" "--workload binop --lt f32 --rt f32 --op add" "--workload binop --lt u8 --rt u8 --op add" "--side 32768 --steps 40"; do
  python bench.py $w --no-cpu-baseline >> $O/bench_all_workloads.jsonl 2>> $O/bench_all.err
done
python tools/expr_cost.py > $O/expr_cost_table.md 2> $O/expr_cost.err
python tools/expr_cost.py --jit > $O/expr_cost_table_compiled.md 2>> $O/expr_cost.err
python tools/kernel_table.py > $O/kernel_table.md 2> $O/kernel_table.err || { tail -20 $O/kernel_table.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
rocprofv3 --kernel-trace --stats -d $O/prof_bench_loop --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-fresh-inputs --no-reference-streams > $O/bench_loop_only_under_rocprof.json 2> $O/prof_bench_loop.err
rocprofv3 --kernel-trace --stats -d $O/prof_evi --output-format csv -- python3 /root/repo/bench.py --workload evi --fused --interpret --no-cpu-baseline > $O/bench_evi_under_rocprof.json 2> $O/prof_evi.err
rocprofv3 --kernel-trace --stats -d $O/prof_evi_compiled --output-format csv -- python3 /root/repo/bench.py --workload evi --fused --no-cpu-baseline > $O/bench_evi_compiled_under_rocprof.json 2> $O/prof_evi_compiled.err
rocprofv3 --kernel-trace --stats -d $O/cov --output-format csv -- python3 -m pytest /root/repo/tests -q -m gpu -p no:cacheprovider \
  -k "not bench and not plain_c and not host_mirror and not config5_example and not duplicate_device and not quick_example" > $O/pytest_under_rocprof.log 2>&1 || { tail -30 $O/pytest_under_rocprof.log; exit 1; }
tail -3 $O/pytest_under_rocprof.log
cd $R
python tools/kernel_coverage.py $O/cov > $O/kernel_instantiation_coverage.md
for d in prof_bench prof_bench_loop prof_evi prof_evi_compiled; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
find $O -name '*.csv' -size +1M -delete
tail -12 $O/kernel_instantiation_coverage.md
python - <<'PY'
import json
for f in ("bench_n1.json", "bench_n1_driver_flags.json", "bench_under_rocprof.json", "bench_loop_only_under_rocprof.json"):
    r = json.load(open("gpurun_out/r03k/" + f)); rf = r["roofline"]
    print(f, round(r["value"], 1), round(rf["frac"], 4), round(rf["launch_ms"], 5), rf.get("fresh_inputs", {}).get("frac"))
for l in open("gpurun_out/r03k/bench_all_workloads.jsonl"):
    r = json.loads(l)
    print(f"{r['config']['workload'][:78]:78s} {r['value']:8.1f} Gcells/s  frac {r['roofline']['frac']:.3f} traffic {r['roofline'].get('traffic')}")
PY

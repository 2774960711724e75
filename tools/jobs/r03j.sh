# Evidence for the compiled form of expression programs: bench line, rocprof kernel stats, cost table, kernel table rows.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
cd $R
python bench.py --workload evi --fused --no-cpu-baseline > $O/bench_evi_compiled.json 2> $O/bench.err
python bench.py --workload evi --fused --interpret --no-cpu-baseline > $O/bench_evi_interpreted.json 2>> $O/bench.err
python bench.py --workload evi --no-cpu-baseline > $O/bench_evi_eager.json 2>> $O/bench.err
python tools/expr_cost.py --jit > $O/expr_cost_table_compiled.md 2> $O/expr_cost.err
python tools/expr_cost.py > $O/expr_cost_table.md 2>> $O/expr_cost.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_evi_compiled --output-format csv -- python3 /root/repo/bench.py --workload evi --fused --no-cpu-baseline > $O/bench_evi_compiled_under_rocprof.json 2> $O/prof.err
cd $R
f=$(find $O/prof_evi_compiled -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_evi_compiled_kernel_stats.csv
find $O -name '*.csv' -size +1M -delete
head -3 $O/bench_evi_compiled_kernel_stats.csv | cut -c1-200
python - <<'PY'
import json
for f in ("bench_evi_compiled.json", "bench_evi_interpreted.json", "bench_evi_eager.json", "bench_evi_compiled_under_rocprof.json"):
    r = json.load(open("gpurun_out/r03j/" + f)); rf = r["roofline"]
    print(f, round(r["value"], 1), round(rf["frac"], 4), round(rf["launch_ms"], 5), rf.get("traffic"), rf["kernel"][:60])
PY

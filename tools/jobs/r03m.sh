# After the host-to-host pipeline: the whole GPU suite, the bench line with --e2e, the host pipeline table.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -9 $O/pytest_gpu.log
python bench.py --e2e > $O/bench_n1_e2e.json 2> $O/bench.err
python tools/host_pipe_bench.py > $O/host_pipe_bench.md 2>> $O/bench.err
cat $O/host_pipe_bench.md
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03m/bench_n1_e2e.json")); rf = r["roofline"]
print(round(r["value"], 1), round(rf["frac"], 4), rf.get("fresh_inputs", {}).get("frac"), r.get("verified"))
for k, v in r.items():
    if "end_to_end" in k:
        print(k, {a: b for a, b in v.items() if a != "what"})
PY

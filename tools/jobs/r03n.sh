# The short divide with its zero-divisor case out of line: parity (the exhaustive operand-space tests included), then the headline.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_instantiations.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "divide or binop or config2" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_a.json 2> $O/err
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_b.json 2>> $O/err
python bench.py --no-cpu-baseline > $O/bench_c.json 2>> $O/err
python bench.py --no-cpu-baseline --workload binop --lt u8 --rt u16 --op add > $O/bench_add.json 2>> $O/err
python bench.py --no-cpu-baseline --workload binop --lt u8 --rt u8 --op div > $O/bench_u8u8div.json 2>> $O/err
python bench.py --no-cpu-baseline --workload binop --lt u8 --rt u8 --op add > $O/bench_u8u8add.json 2>> $O/err
python bench.py --no-cpu-baseline --rows 2048 > $O/bench_shard.json 2>> $O/err
python - <<'PY'
import json
for f in ("bench_a", "bench_b", "bench_c", "bench_add", "bench_u8u8div", "bench_u8u8add", "bench_shard"):
    r = json.load(open(f"gpurun_out/r03n/{f}.json")); rf = r["roofline"]
    print(f, round(r["value"], 1), round(rf["frac"], 4), round(rf["launch_ms"], 5), rf.get("fresh_inputs", {}).get("frac"), r.get("verified"))
PY

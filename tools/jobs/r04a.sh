# Round 4, first call: the rotating-set headline (bench.py), its contract tests, the store-stream sweep.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
python bench.py > $O/bench_n1.json 2> $O/err || { tail -30 $O/err; exit 1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1_driver_flags.json 2>> $O/err
python bench.py --no-cpu-baseline --no-reference-streams --tune mall_mb=0 > $O/bench_n1_all_nt.json 2>> $O/err
python bench.py --no-cpu-baseline --no-reference-streams --tune mall_mb=192 > $O/bench_n1_mall192.json 2>> $O/err
python bench.py --no-cpu-baseline --no-reference-streams --rows 2048 > $O/bench_shard8.json 2>> $O/err
python bench.py --no-cpu-baseline --no-reference-streams --rows 2048 --tune mall_mb=0 > $O/bench_shard8_all_nt.json 2>> $O/err
python bench.py --no-cpu-baseline --no-reference-streams --sets 1 > $O/bench_n1_sets1.json 2>> $O/err
python - <<'PY'
import json
for f in ("bench_n1", "bench_n1_driver_flags", "bench_n1_all_nt", "bench_n1_mall192", "bench_shard8", "bench_shard8_all_nt", "bench_n1_sets1"):
    r = json.load(open(f"gpurun_out/r04a/{f}.json")); rf = r["roofline"]
    print(f, round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5), "sets", r["config"]["operand_sets"],
          "resident", round(rf.get("cache_resident_loop", {}).get("frac", 0), 4), r.get("verified"), rf.get("reference_streams"))
PY
./tools/tune_store 16384 9 10 > $O/tune_store.log 2> $O/tune_store.err || { tail -5 $O/tune_store.err; }
sort -k7 -n -r -t$'\t' $O/tune_store.log | head -3 > /dev/null
grep -E "^(wr |mix|ref)" $O/tune_store.log | sort -t'|' -k1,1 | awk '{print}' | sort -k1,1 -s | head -0
python - <<'PY'
rows = [l.rstrip("\n") for l in open("gpurun_out/r04a/tune_store.log") if l[:3] in ("wr ", "mix", "ref")]
for kind in ("wr ", "mix", "ref"):
    sel = sorted((l for l in rows if l.startswith(kind)), key=lambda l: -float(l.split()[-2]))
    print("\n".join(sel[:8])); print("   ... shipped:", next((l for l in sel if "U2 x4w wg-interleave nt 2fronts lds0K" in l), "")); print()
PY
python -m pytest tests/test_bench_contract.py -x -q -m gpu > $O/pytest_bench.log 2>&1 || { tail -40 $O/pytest_bench.log; exit 1; }
tail -3 $O/pytest_bench.log

# Round 4, second call: write-through stores (EC_STORE_POLICY 2 = `sc1 nt`, the library's default now) against `nt` alone
# (liberased_cells_hip_ntstore.so), the whole GPU suite on the new stores, the second store-stream sweep.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -30; }
tail -3 $O/pytest_gpu.log
NT=$R/erased-cells_amd/liberased_cells_hip_ntstore.so
for rep in 1 2; do
python bench.py --no-cpu-baseline > $O/bench_sc1_$rep.json 2>> $O/err
EC_HIP_LIB=$NT python bench.py --no-cpu-baseline > $O/bench_nt_$rep.json 2>> $O/err
done
python bench.py --no-cpu-baseline --rows 2048 > $O/bench_shard8_sc1.json 2>> $O/err
EC_HIP_LIB=$NT python bench.py --no-cpu-baseline --rows 2048 > $O/bench_shard8_nt.json 2>> $O/err
python bench.py --no-cpu-baseline --no-reference-streams --tune mall_mb=0 > $O/bench_sc1_all_nt_loads.json 2>> $O/err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04b/bench_*.json")):
    r = json.load(open(f)); rf = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5), "sets", r["config"]["operand_sets"],
          "resident", round(rf.get("cache_resident_loop", {}).get("frac", 0), 4), r.get("verified"),
          {k: round(v) for k, v in rf.get("reference_streams", {}).items() if k != "what"})
PY
./tools/tune_store 16384 9 10 > $O/tune_store_v2.log 2> $O/tune_store_v2.err || { tail -5 $O/tune_store_v2.err; }
python - <<'PY'
import re
rows = [l.rstrip("\n") for l in open("gpurun_out/r04b/tune_store_v2.log") if l[:3] in ("wr ", "mix", "ref", "rd1")]
key = lambda l: -float(re.search(r"(\d\.\d+)\s+\S+$", l).group(1))
for kind in ("wr ", "mix", "rd1", "ref"):
    print("\n".join(sorted((l for l in rows if l.startswith(kind)), key=key))); print()
PY
python tools/kernel_table.py > $O/kernel_table_sc1.md 2>> $O/err
EC_HIP_LIB=$NT python tools/kernel_table.py > $O/kernel_table_nt.md 2>> $O/err
python - <<'PY'
def rows(f):
    d = {}
    for l in open(f):
        c = [x.strip() for x in l.split("|")]
        if len(c) > 8 and c[1] and c[1] != "kernel (through the C ABI)" and not c[1].startswith("---"):
            d[c[1]] = (float(c[6]), float(c[8]))
    return d
a, b = rows("gpurun_out/r04b/kernel_table_sc1.md"), rows("gpurun_out/r04b/kernel_table_nt.md")
print("%-70s %8s %8s   %8s %8s" % ("kernel", "sc1 loop", "nt loop", "sc1 HBM", "nt HBM"))
for k in a:
    if k in b: print("%-70s %8.3f %8.3f   %8.3f %8.3f" % (k[:70], a[k][0], b[k][0], a[k][1], b[k][1]))
PY

# Round 4: store policy decided per stream kind (EC_STORE_POLICY 3: `sc1 nt` for value streams, `nt` for mask streams) against
# `sc1 nt` everywhere (…_sc1all.so) and `nt` everywhere (…_ntstore.so), interleaved twice; the staged short divide; parity on the new stores.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_instantiations.py tests/test_gpu_fullsize.py tests/test_abi_host.py -q -m gpu > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -30; }
tail -2 $O/pytest_gpu.log
A=$R/erased-cells_amd/liberased_cells_hip_sc1all.so; N=$R/erased-cells_amd/liberased_cells_hip_ntstore.so
for rep in 1 2; do
  python bench.py --no-cpu-baseline > $O/bench_p3_$rep.json 2>> $O/err
  EC_HIP_LIB=$A python bench.py --no-cpu-baseline > $O/bench_sc1all_$rep.json 2>> $O/err
  EC_HIP_LIB=$N python bench.py --no-cpu-baseline > $O/bench_nt_$rep.json 2>> $O/err
  python tools/kernel_table.py > $O/kernel_table_p3_$rep.md 2>> $O/err
  EC_HIP_LIB=$A python tools/kernel_table.py > $O/kernel_table_sc1all_$rep.md 2>> $O/err
  EC_HIP_LIB=$N python tools/kernel_table.py > $O/kernel_table_nt_$rep.md 2>> $O/err
done
python bench.py --no-cpu-baseline --rows 2048 > $O/bench_shard8_p3.json 2>> $O/err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04d/bench_*.json")):
    r = json.load(open(f)); rf = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 1), "frac", round(rf["frac"], 4), "ms", round(rf["launch_ms"], 5),
          "resident", round(rf.get("cache_resident_loop", {}).get("frac", 0), 4), r.get("verified"),
          {k: round(v) for k, v in rf.get("reference_streams", {}).items() if k != "what"})
def rows(f):
    d = {}
    for l in open(f):
        c = [x.strip() for x in l.split("|")]
        if len(c) > 8 and c[1] and c[1] != "kernel (through the C ABI)" and not c[1].startswith("---"):
            d[c[1]] = (float(c[6]), float(c[8]))
    return d
T = {k: [rows(f"gpurun_out/r04d/kernel_table_{k}_{r}.md") for r in (1, 2)] for k in ("p3", "sc1all", "nt")}
print("\nall-HBM column (every load nt), two runs each, and the one-set loop column\n%-66s %13s %13s %13s | %6s %6s %6s" % ("kernel", "values-sc1", "sc1 all", "nt all", "loop", "loop", "loop"))
for k in T["p3"][0]:
    f = lambda name, col: "/".join(f"{t[k][col]:.3f}" for t in T[name] if k in t)
    m = lambda name, col: sum(t[k][col] for t in T[name] if k in t) / max(1, sum(1 for t in T[name] if k in t))
    print("%-66s %13s %13s %13s | %6.3f %6.3f %6.3f" % (k[:66], f("p3", 1), f("sc1all", 1), f("nt", 1), m("p3", 0), m("sc1all", 0), m("nt", 0)))
PY

# Round 4: which operand streams are worth loading with the default cache policy (cache_plan, csrc/ec_runtime.hpp)?  Every binop shape
# of interest under each forced policy (bit 0 = lhs, bit 1 = rhs cacheable): the rotating-set rate (nothing is ever re-read: what the
# policy COSTS) and the one-set loop (the operand is re-read by the next launch: what it GAINS).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
for wl in "u8 u16 div" "u8 u16 add" "u8 u8 div" "u8 u8 add" "u16 u8 add" "u16 u16 add"; do
  set -- $wl
  for f in 0 1 2 3; do
    python bench.py --no-cpu-baseline --workload binop --lt $1 --rt $2 --op $3 --tune cache_force=$f --steps 100 > $O/binop_$1_$2_$3_force$f.json 2>> $O/err
  done
  python bench.py --no-cpu-baseline --workload binop --lt $1 --rt $2 --op $3 --steps 100 > $O/binop_$1_$2_$3_policy.json 2>> $O/err
done
python - <<'PY'
import json, glob
print("| workload | policy | rotating sets (all-HBM) frac | one-set loop frac |\n|---|---|---:|---:|")
for f in sorted(glob.glob("gpurun_out/r04c/binop_*.json")):
    r = json.load(open(f)); rf = r["roofline"]
    name = f.split("/")[-1][6:-5]
    wl, pol = name.rsplit("_", 1)
    print(f"| {wl} | {pol} | {rf['frac']:.4f} | {rf['cache_resident_loop']['frac']:.4f} |")
PY

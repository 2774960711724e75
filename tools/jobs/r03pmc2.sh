# Counters for the expression-program kernel: HBM bytes of the EVI workload (one pass and eager), and the instruction mix /
# issue utilisation of k_expr that the cost model in DESIGN §5 rests on.  Same rules as r03pmc.sh.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 5 --warmup 1 --ramp 0 --no-cpu-baseline --no-reference-streams --no-fresh-inputs"
run() {  # name, counters, bench flags
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs --kernel-trace -d $O/$name --output-format csv -- python3 /root/repo/bench.py $COMMON "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  find $O/$name -name '*kernel_trace.csv' -delete
  echo "done $name"
}
for wl in "evi_fused --workload evi --fused --interpret" "evi_fused_compiled --workload evi --fused" "evi --workload evi"; do
  set -- $wl; key=$1; shift
  run ${key}__fetch FETCH_SIZE "$@"
  run ${key}__write WRITE_SIZE "$@"
done
run evi_fused__insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" --workload evi --fused --interpret
run evi_fused__active "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --workload evi --fused --interpret
run evi_fused__wait "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" --workload evi --fused --interpret
run evi_fused_compiled__insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" --workload evi --fused
run evi_fused_compiled__active "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --workload evi --fused
run ndvi_fused__insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" --workload ndvi --fused
run ndvi_fused__active "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --workload ndvi --fused
cd $R
python tools/pmc_summary.py $O --steps 6 --out-json $O/pmc_summary.json --out-md $O/pmc_summary.md
cat $O/pmc_summary.md | head -20
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03pmc2/*__insts") + glob.glob("gpurun_out/r03pmc2/*__active") + glob.glob("gpurun_out/r03pmc2/*__wait")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_expr" in r["Kernel_Name"] or "k_fused_any" in r["Kernel_Name"] or "ec_expr_jit" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        vals = list(v.values())
        print(f"{d.split('/')[-1]:24s} {k:24s} dispatches {len(vals):3d}  mean {sum(vals)/len(vals):.5g}")
PY
du -sh $O

# Round 4: memory-side counters of the store stream for the shipped shape and the shapes that beat it (tools/tune_store.hip, `only`
# form: the program directly after `rocprofv3 … --`, one counter group per pass, --kernel-trace only).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04store_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, counters, variant name
  local tag=$1 ctrs=$2 name=$3
  rocprofv3 --pmc $ctrs --kernel-trace -d $O/$tag --output-format csv -- /root/repo/tools/tune_store 16384 3 10 "$name" > $O/$tag.log 2> $O/$tag.err || { tail -5 $O/$tag.err; return 1; }
  find $O/$tag -name '*kernel_trace.csv' -delete
}
i=0
while IFS='|' read -r key name; do
  i=$((i+1))
  run ${key}__wr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "$name"
  run ${key}__rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" "$name"
  run ${key}__st "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_WRITE_sum" "$name"
  echo "done $key"
done <<'LIST'
wr_shipped|wr  U2 x4w wg-interleave nt 2fronts lds0K
wr_nt_2percu|wr  U2 x4w wg-interleave nt 2fronts lds64K
wr_sc1_full|wr  U2 x4w wg-interleave nt+sc1 2fronts lds0K
wr_sc1_2percu|wr  U2 x4w wg-interleave nt+sc1 2fronts lds64K
wr_nt_1percu|wr  U2 x4w wg-interleave nt 2fronts lds96K
mix_shipped|mix U2 x4w wg-interleave nt 2fronts lds0K
mix_sc1|mix U2 x4w wg-interleave nt+sc1 2fronts lds0K
mix_sc1_6percu|mix U2 x4w wg-interleave nt+sc1 2fronts lds24K
LIST
cd $R
python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r04store_pmc"
rows = collections.OrderedDict()
for d in sorted(glob.glob(O + "/*__*")):
    if not os.path.isdir(d): continue
    key = os.path.basename(d).split("__")[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_store" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            v = v[len(v) // 3:]  # drop the ramp
            rows.setdefault(key, {})[c] = sum(v) / len(v)
cols = ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_STALL_sum",
        "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_NORMAL_WRITEBACK_sum", "TCC_WRITE_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum",
        "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "GRBM_GUI_ACTIVE"]
print("| variant | " + " | ".join(c.replace("TCC_", "").replace("_sum", "") for c in cols) + " |")
print("|---|" + "---:|" * len(cols))
for k, d in rows.items():
    print(f"| {k} | " + " | ".join(f"{d.get(c, float('nan')):.4g}" for c in cols) + " |")
PY
du -sh $O

# Round 4: HBM traffic of every bench workload again, on the kernels as they ship now (write-through value stores, the built-in
# expression kernels): FETCH_SIZE / WRITE_SIZE in separate passes, the request / level / stall set for the headline and the store-heavy
# kernels.  One rocprofv3 pass per counter group, the program directly after `--`, --kernel-trace only.  tools/pmc_summary.py.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 5 --warmup 1 --ramp 0 --no-cpu-baseline --no-reference-streams --no-resident-loop"
run() {  # name, counters, bench flags
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs --kernel-trace -d $O/$name --output-format csv -- python3 /root/repo/bench.py $COMMON "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  find $O/$name -name '*kernel_trace.csv' -delete
  echo "done $name"
}
for wl in "div" "masked_chain --workload masked_chain" "masked_chain_fused --workload masked_chain --fused" "minmax --workload minmax" \
          "ndvi_fused --workload ndvi --fused" "ndvi_fused_mixed --workload ndvi --fused --mixed" "binop_add_u16_u16 --workload binop --lt u16 --rt u16 --op add" \
          "binop_add_f32_f32 --workload binop --lt f32 --rt f32 --op add" "evi_fused_builtin --workload evi --fused" "evi_fused --workload evi --fused --interpret" \
          "evi_fused_compiled --workload evi --fused --compiled" "evi --workload evi"; do
  set -- $wl; key=$1; shift
  run ${key}__fetch FETCH_SIZE "$@"
  run ${key}__write WRITE_SIZE "$@"
done
for wl in "div" "ndvi_fused --workload ndvi --fused" "binop_add_f32_f32 --workload binop --lt f32 --rt f32 --op add" "evi_fused_builtin --workload evi --fused"; do
  set -- $wl; key=$1; shift
  run ${key}__req "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "$@"
  run ${key}__lvl "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "$@"
  run ${key}__hit "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "$@"
  run ${key}__stall "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUBBLE_sum TCC_NORMAL_WRITEBACK_sum TCC_EA0_RDREQ_32B_sum" "$@"
done
cd $R
python tools/pmc_summary.py $O --steps 6 --out-json $O/pmc_summary.json --out-md $O/pmc_summary.md --traffic $O/traffic.json --commit ${EC_COMMIT:-r04} --round 4
head -20 $O/pmc_summary.md
du -sh $O

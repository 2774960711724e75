# Round 4: the LDS-staged binop kernel as launched by rule against the direct kernel on the same operands (f64 + u16, 18 B/cell): traffic and the
# request / level / stall counters.  One rocprofv3 pass per counter group, the program directly after `--`, --kernel-trace only.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 5 --warmup 1 --ramp 0 --no-cpu-baseline --no-reference-streams --no-resident-loop --workload binop --lt f64 --rt u16 --op add"
run() {
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs --kernel-trace -d $O/$name --output-format csv -- python3 /root/repo/bench.py $COMMON "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  find $O/$name -name '*kernel_trace.csv' -delete
  echo "done $name"
}
for v in "rule -1" "direct 0"; do
  set -- $v; key=binop_add_f64_u16_$1; var=$2
  run ${key}__fetch FETCH_SIZE --tune binop_variant=$var
  run ${key}__write WRITE_SIZE --tune binop_variant=$var
  run ${key}__req "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" --tune binop_variant=$var
  run ${key}__lvl "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" --tune binop_variant=$var
  run ${key}__stall "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUBBLE_sum TCC_NORMAL_WRITEBACK_sum TCC_EA0_RDREQ_32B_sum" --tune binop_variant=$var
done
cd $R
python tools/pmc_summary.py $O --steps 6 --out-json $O/pmc_summary.json --out-md $O/pmc_summary.md
cat $O/pmc_summary.md | cut -c1-250

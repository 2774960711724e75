set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z; mkdir -p $O
cd $R
for m in default uncached finegrained; do
  rc=0
  timeout -k 10 200 ./tools/tune_store6 16384 7 10 "" $m > $O/tune_store_v6_$m.log 2> $O/err_$m || rc=$?
  if grep -q "Memory access fault" $O/err_$m; then echo FAULT $m; exit 9; fi
  echo "== $m rc $rc"; grep -v "^rd1" $O/tune_store_v6_$m.log
done

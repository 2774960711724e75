set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z3; mkdir -p $O
cd $R
rc=0
timeout -k 10 300 ./tools/tune_store8 16384 7 10 > $O/tune_store_v8.log 2> $O/err || rc=$?
if grep -q "Memory access fault" $O/err; then echo FAULT; exit 9; fi
grep -v "^rd1" $O/tune_store_v8.log
exit $rc

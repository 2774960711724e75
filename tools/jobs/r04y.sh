set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04y; mkdir -p $O
cd $R
rc=0
timeout -k 10 420 ./tools/tune_store5 16384 7 10 > $O/tune_store_v5.log 2> $O/err || rc=$?
tail -3 $O/err
cat $O/tune_store_v5.log
if grep -q "Memory access fault" $O/err; then exit 9; fi
exit $rc

set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout -k 10 400 ./tools/tune_nt_width 16384 9 > $O/tune_nt_width_rotating.log 2>&1 || { tail -20 $O/tune_nt_width_rotating.log; exit 1; }
cat $O/tune_nt_width_rotating.log

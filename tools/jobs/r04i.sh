R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
timeout -k 10 120 python tools/probe_unmap_reuse.py refuse 40 > $O/refuse.log 2>&1; echo "refuse rc $?"; tail -4 $O/refuse.log | cut -c1-300
timeout -k 10 120 python tools/probe_unmap_reuse.py real 40 > $O/real.log 2>&1; echo "real rc $?"; tail -4 $O/real.log | cut -c1-300

# the whole GPU suite once more, uncaptured, so that a runtime message in front of an abort is kept
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -q -m gpu -s > $O/stdout.log 2> $O/stderr.log; echo "rc $?"
tail -3 $O/stdout.log | cut -c1-300; grep -v "^Extension" $O/stderr.log | tail -15 | cut -c1-300

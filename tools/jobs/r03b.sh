# Round-3 second run: the reworked shard group (tests + host cost), nt policy per load width, fused tile depth.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_sharded_group.py tests/test_gpu_instantiations.py -x -q -m gpu --durations=5 -k "not exact_for_every" > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -12 $O/pytest_gpu.log
timeout -k 10 400 ./tools/tune_nt_width 16384 9 > $O/tune_nt_width.log 2>&1 || { tail -20 $O/tune_nt_width.log; exit 1; }
cat $O/tune_nt_width.log
timeout -k 10 300 ./tools/tune_fused_any_u1 16384 7 > $O/tune_fused_any_u1.log 2>&1 || { tail -20 $O/tune_fused_any_u1.log; exit 1; }
tail -14 $O/tune_fused_any_u1.log
echo "| G | cells/shard | K | async host/wall/gpu us | blocking host/wall/gpu us | plain one thread host/wall/gpu us | async again |" > $O/group_fanout_rows.md
for G in 1 2 4 8; do
  timeout -k 10 120 ./tools/group_bench $G 65536 200 >> $O/group_fanout_rows.md 2>> $O/group_bench.err
  timeout -k 10 120 ./tools/group_bench $G $((33554432 / G)) 200 >> $O/group_fanout_rows.md 2>> $O/group_bench.err
done
timeout -k 10 120 ./tools/group_bench 8 33554432 100 >> $O/group_fanout_rows.md 2>> $O/group_bench.err
cat $O/group_fanout_rows.md

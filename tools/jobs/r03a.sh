# Round-3 first run: GPU suite on the reworked loads (1-byte cells as words, nt everywhere) and k_fused_any, then the
# two A/B tuners, then the headline bench.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu --durations=10 > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -16 $O/pytest_gpu.log
timeout -k 10 300 ./tools/tune_nt_u8 16384 11 > $O/tune_nt_u8.log 2>&1 || { tail -20 $O/tune_nt_u8.log; exit 1; }
tail -18 $O/tune_nt_u8.log
timeout -k 10 300 ./tools/tune_fused_any 16384 9 > $O/tune_fused_any.log 2>&1 || { tail -20 $O/tune_fused_any.log; exit 1; }
cat $O/tune_fused_any.log
timeout -k 10 300 ./tools/tune_fused_any_u2 16384 9 > $O/tune_fused_any_u2.log 2>&1 || { tail -20 $O/tune_fused_any_u2.log; exit 1; }
tail -14 $O/tune_fused_any_u2.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cat $O/bench_n1.json

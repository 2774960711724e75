set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02e; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python tools/kernel_table.py > $O/kernel_table.md 2> $O/kernel_table.err
grep -E "min_max|mask_counts" $O/kernel_table.md
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cut -c1-300 $O/bench_n1.json

set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04w; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_kats.py tests/test_gpu_fullsize.py -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault|Error" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python tools/kernel_table.py > $O/kernel_table.md 2> $O/err
grep -E "^\| (first_diff|min_max UInt8|mask_counts)" $O/kernel_table.md | cut -c1-140

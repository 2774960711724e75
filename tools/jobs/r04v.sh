set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04v; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_instantiations.py tests/test_gpu_reference_kats.py -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault|Error" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python tools/write_heavy_caps.py > $O/write_heavy_caps.md 2> $O/err; head -7 $O/write_heavy_caps.md
python tools/kernel_table.py > $O/kernel_table.md 2>> $O/err
grep -E "^\| (binop_scalar)" $O/kernel_table.md | cut -c1-140

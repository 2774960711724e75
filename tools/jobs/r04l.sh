# Round 4: stream placement probe; the kernel table at 32768² (the reductions against raster size); synth generators without the cap.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_kats.py -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python tools/placement_probe.py > $O/placement_probe.md 2> $O/err || tail -5 $O/err
cat $O/placement_probe.md
python tools/kernel_table.py 32768 > $O/kernel_table_32768.md 2>> $O/err || tail -5 $O/err
grep -E "^\| (min_max|mask_counts|first_diff|fill|binop (Add|Div) UInt8)" $O/kernel_table_32768.md | cut -c1-150
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2>> $O/err
python -c "
import json; r=json.load(open('gpurun_out/r04l/bench.json')); print(round(r['value'],1), round(r['roofline']['frac'],4), r['verified'])"

# end-of-round check of the library as it ships: the whole GPU suite, smoke(), the kernel table, the headline.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04r; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || tail -5 $O/smoke.log
tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2> $O/err
python tools/kernel_table.py > $O/kernel_table.md 2>> $O/err
python -c "
import json; r=json.load(open('gpurun_out/r04r/bench_n1_driver_flags.json')); print(round(r['value'],1), round(r['roofline']['frac'],4), r['verified'], round(r['roofline']['cache_resident_loop']['frac'],4))"
grep -E "^\| (mask_counts|min_max UInt8 \||binop Div UInt8∘UInt16|fill)" $O/kernel_table.md | cut -c1-150

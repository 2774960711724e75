# Round-2 evidence run: tests, bench lines, rocprofv3 kernel stats, kernel table.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -12 $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for w in "--rows 2048" "--rows 2048 --graph" "--workload masked_chain" "--workload masked_chain --fused" "--workload minmax" "--workload minmax --side 32768" "--workload minmax --side 65536 --steps 20 --warmup 3" "--workload ndvi" "--workload ndvi --fused" "--workload ndvi --mixed" "--workload ndvi --fused --mixed"; do
  python bench.py $w --no-cpu-baseline >> $O/bench_all_workloads.jsonl 2>> $O/bench_all.err
done
python tools/reduce_shape_ab.py > $O/reduce_shape_ab.md 2>> $O/bench_all.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
rocprofv3 --kernel-trace --stats -d $O/prof_table --output-format csv -- python3 /root/repo/tools/kernel_table.py > $O/kernel_table.md 2> $O/prof_table.err
cd $R
for d in prof_bench prof_table; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
  find $O/$d -name '*.csv' -size +1M -delete
done
head -5 $O/prof_bench_kernel_stats.csv
cat $O/kernel_table.md | tail -50

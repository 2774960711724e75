# Round-3: counter evidence (r03pmc.sh) and the kernel table, plain and under rocprofv3, on the load-policy build.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
python tools/kernel_table.py > $O/kernel_table.md 2> $O/kernel_table.err || { tail -20 $O/kernel_table.err; exit 1; }
tail -60 $O/kernel_table.md
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_table --output-format csv -- python3 /root/repo/tools/kernel_table.py > $O/kernel_table_under_rocprof.md 2> $O/prof_table.err
rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
cd $R
for d in prof_bench prof_table; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
  find $O/$d -name '*.csv' -size +1M -delete
done
head -6 $O/prof_bench_kernel_stats.csv
bash tools/jobs/r03pmc.sh > $O/pmc.log 2>&1 || { tail -30 $O/pmc.log; exit 1; }
tail -30 $O/pmc.log

set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O
cd $R
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2> $O/bench_n1_driver.err
python bench.py --gpus 2 --backend gloo --single-device --steps 20 --warmup 5 > $O/bench_gloo2_one_device.json 2> $O/bench_gloo2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
cd $R
f=$(find $O/prof_bench -name '*kernel_stats.csv' | head -1); cp $f $O/prof_bench_kernel_stats.csv
find $O/prof_bench -name '*.csv' -size +1M -delete
python - <<'PY'
import json
for f in ("bench_n1","bench_n1_driver_flags","bench_gloo2_one_device","bench_under_rocprof"):
    d=json.load(open(f"gpurun_out/r02s/{f}.json")); print(f, d["n_gpus"], round(d["value"],1), round(d["ms_per_step"],4), round(d["roofline"]["launch_ms"],4), round(d["roofline"]["frac"],4), d.get("verified"))
PY
head -2 $O/prof_bench_kernel_stats.csv | cut -c1-300

set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
python -m pytest tests/test_bench_contract.py -x -q -m gpu > $O/pytest_bench.log 2>&1 || { tail -30 $O/pytest_bench.log; exit 1; }
tail -3 $O/pytest_bench.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --workload minmax --side 65536 --steps 20 --warmup 3 > $O/bench_minmax_65536.json 2> $O/bench_minmax_65536.err
python bench.py --workload minmax --side 32768 --steps 50 --warmup 5 > $O/bench_minmax_32768.json 2> $O/bench_minmax_32768.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 1 --ramp 0 --no-cpu-baseline --no-reference-streams > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 1 --ramp 0 --no-cpu-baseline --no-reference-streams > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --summary $O/pmc_summary.json --traffic $O/traffic.json > $O/pmc_traffic.out
# keep only the small CSVs
find $O/pmc_fetch $O/pmc_write -name '*.csv' -size +2M -delete
cat $O/bench_n1.json; cat $O/bench_minmax_65536.json

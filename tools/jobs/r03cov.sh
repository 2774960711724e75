set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03cov; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/cov --output-format csv -- python3 -m pytest /root/repo/tests -q -m gpu -p no:cacheprovider \
  -k "not bench and not plain_c and not host_mirror and not config5_example and not duplicate_device and not quick_example" > $O/pytest_under_rocprof.log 2>&1 || { tail -30 $O/pytest_under_rocprof.log; exit 1; }
tail -3 $O/pytest_under_rocprof.log
cd $R
python tools/kernel_coverage.py $O/cov > $O/kernel_instantiation_coverage.md
find $O/cov -name '*.csv' -size +1M -delete
cat $O/kernel_instantiation_coverage.md

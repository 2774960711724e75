# debug: the abort in tests/test_gpu_soak.py behind the full suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
which gdb > $O/which_gdb.txt 2>&1
echo "== soak alone"; timeout -k 10 200 python -m pytest tests/test_gpu_soak.py -q -m gpu -x > $O/soak_alone.log 2>&1; echo "rc $?"; tail -3 $O/soak_alone.log | cut -c1-300
echo "== refused-pin test + soak"; timeout -k 10 300 python -m pytest tests/test_gpu_sharded_group.py::test_host_to_host_when_page_locking_is_refused tests/test_gpu_soak.py -q -m gpu -x > $O/pin_then_soak.log 2>&1; echo "rc $?"; grep -v "^Extension" $O/pin_then_soak.log | tail -5 | cut -c1-300
echo "== sharded group + soak"; timeout -k 10 400 python -m pytest tests/test_gpu_sharded_group.py tests/test_gpu_soak.py -q -m gpu -x > $O/group_then_soak.log 2>&1; echo "rc $?"; grep -v "^Extension" $O/group_then_soak.log | tail -5 | cut -c1-300
echo "== reference kats + soak"; timeout -k 10 400 python -m pytest tests/test_gpu_reference_kats.py tests/test_gpu_soak.py -q -m gpu -x > $O/kats_then_soak.log 2>&1; echo "rc $?"; grep -v "^Extension" $O/kats_then_soak.log | tail -5 | cut -c1-300

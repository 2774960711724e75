# Mask::counts in one launch (packed ticket + sum atomic) against partials + finalize: parity, then the rate at 16384² and 4096².
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_kats.py tests/test_gpu_fullsize.py -q -m gpu -s > $O/pytest_gpu.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Abort|fault|Error" $O/pytest_gpu.log | tail -30; }
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python - <<'PY' | tee $O/counts_ab.txt
import sys, ctypes as C
sys.path.insert(0, "erased-cells_amd/python")
import torch, erased_cells_hip as ec
torch.cuda.set_device(0); ec.init(0); L = ec.lib(); chk = ec._ffi.check
s = torch.cuda.current_stream().cuda_stream
for side in (16384, 4096, 32768):
    n = side * side
    ms = [ec.Mask.empty(n) for _ in range(4 if side <= 16384 else 2)]
    for k, m in enumerate(ms): chk(L.ec_synth_mask(m.mem.ptr, n, 100 + k, 0, 30, s))
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    for rep in range(2):
        for mode in (0, 1):
            chk(L.ec_tune_set(b"counts_one_launch", mode))
            def run(k):
                for i in range(k): chk(L.ec_mask_counts_device(ms[i % len(ms)].mem.ptr, n, out.data_ptr(), s))
            run(300); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(600); e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 600
            print(f"mask_counts {side}^2 rotating {len(ms)} masks, counts_one_launch={mode}: {t * 1e3:.2f} us, {n / t / 1e6 / 8000:.4f} of peak, result {out.tolist()}")
PY

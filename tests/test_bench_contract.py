"""bench.py's output contract: ONE JSON line on stdout with BASELINE.json's metric plus the
`roofline` and `cpu_baseline` objects; and a loud failure (no fallback) without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_bench_emits_one_json_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, BENCH, "--side", "4096", "--steps", "10", "--warmup", "2", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "Gcells/s"
    for k, t in (("value", float), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["verified"] is True and d["cpu_baseline"]["oracle_check_of_timed_output"] is True
    assert d["roofline"]["traffic"] is None or "recorded profiles/traffic.json" in d["roofline"]["traffic_source"]
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["cells"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - 11 * rf["cells_per_launch"] / (rf["launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert 0.05 < rf["frac"] < 1.0
    # the headline is an all-HBM figure: the timed steps rotate over >= 4 operand sets with >= 1 GiB between two uses of a byte;
    # the one-set loop of rounds 1-3 is a side figure, for every rank
    cfg = d["config"]
    assert rf["all_hbm"] is True and cfg["operand_sets"] >= 4 and "rotation" in cfg
    assert (cfg["operand_sets"] - 1) * cfg["operand_set_bytes_per_rank"] >= 1 << 30
    assert cfg["operand_set_bytes_per_rank"] == 11 * rf["cells_per_launch"]
    assert cfg["untimed_steps_before_timing"] == cfg["clock_ramp_steps"] + d["warmup"]
    loop = rf["cache_resident_loop"]
    assert loop["operand_sets"] == 1 and 0.05 < loop["frac"] and "NOT an HBM figure" in loop["what"]
    assert abs(loop["frac"] - 11 * rf["cells_per_launch"] / (loop["launch_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    assert rf["per_gpu"][0]["cache_resident_loop"]["launch_ms"] == loop["launch_ms"] and "fresh_inputs" not in rf
    rs = rf["reference_streams"]  # untimed plain streams of the same buffers (SURVEY §8d "empirical ceiling")
    assert all(0 < rs[k] < 8000.0 for k in ("same_mix_add_u8_u16_GBps", "write_only_fill_f64_GBps", "read_only_min_max_f64_GBps"))
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Gcells/s" and cb["value"] > 0 and cb["sample"]


def _one_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus_n_launches_itself_and_names_the_device_count():
    """`python bench.py --gpus 2` with no launcher around it must start its own ranks; on a box with fewer than
    two GPUs it fails loudly, naming how many it saw, and never asks for torchrun."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two HIP devices present: the launch would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert f"shows {torch.cuda.device_count()} HIP device(s)" in r.stderr and "torch.distributed.run" not in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_self_launched_two_ranks_rehearsal_on_one_gpu():
    """The N>1 control flow end to end from a bare `python bench.py --gpus 2`: self-launch of two fresh ranks,
    row-block shards, per-rank gather, MAX-over-ranks time, one JSON line.  Both ranks share device 0 and talk
    over gloo (RCCL refuses two ranks on one GPU) — a rehearsal of the flow, not a scaling measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--single-device", "--side", "4096",
                        "--steps", "5", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=800)
    d = _one_line(r)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["verified"] is True
    assert d["config"]["cells_per_rank"] == [4096 * 2048, 4096 * 2048] and d["config"]["cells"] == 4096 * 4096
    pg = d["roofline"]["per_gpu"]
    assert [g["rank"] for g in pg] == [0, 1] and all(g["cells"] == 4096 * 2048 and 0 < g["frac"] < 1 for g in pg)
    assert abs(d["roofline"]["launch_ms"] - max(g["launch_ms"] for g in pg)) < 1e-12
    # every rank rotates its own shard's operand sets and reports both figures
    assert d["roofline"]["all_hbm"] is True and d["config"]["operand_sets"] >= 4
    assert all(g["cache_resident_loop"]["frac"] > 0 for g in pg)  # may exceed 1: a 92 MB set lives in the Infinity Cache — not an HBM figure
    assert abs(d["roofline"]["cache_resident_loop"]["launch_ms"] - max(g["cache_resident_loop"]["launch_ms"] for g in pg)) < 1e-12
    assert "cpu_baseline" not in d  # rank 0 at N=1 only


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_rccl_path_with_one_rank():
    """backend nccl (= RCCL) through the same distributed code path, with the one rank a 1-GPU box allows."""
    env = dict(os.environ, EC_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--side", "4096", "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                        "--workload", "minmax"], env=env, capture_output=True, text=True, timeout=500)
    d = _one_line(r)
    assert d["n_gpus"] == 1 and d["dtype"] == "u16" and len(d["roofline"]["per_gpu"]) == 1

"""Multi-rank restoration on the GPU box: two processes (torch.distributed.run) share the one GPU, gloo carries the
weight broadcast / gather / reductions (RCCL wants one GPU per rank; the collective calls are the same)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_job_reproduces_the_single_process_result():
    env = dict(os.environ, DDPMIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DIST-GPU-OK images=5" in r.stdout


@pytest.mark.gpu
def test_rccl_loopback():
    """RCCL itself, as far as one GPU allows: a world-size-1 "nccl" process group runs every collective of dist.py with its
    dtypes and shapes (tests/rccl_loopback_worker.py).  The multi-GPU curve is the driver's to measure."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_loopback_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL-LOOPBACK-OK" in r.stdout


@pytest.mark.gpu
def test_bench_contract_two_ranks():
    """bench.py under the driver's launch line (2 ranks): one JSON line from rank 0, whole-job value, no cpu_baseline."""
    import json
    env = dict(os.environ, DDPMIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29542", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--size", "64"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 4 and "cpu_baseline" not in d
    assert abs(d["value"] - 4 / (1000 * d["ms_per_step"] / 1e3)) < 1e-9


@pytest.mark.gpu
def test_bench_contract_single_gpu():
    """`python bench.py` (N=1) on the DEFAULT workload (configs[1]; two of the extra slices to keep it short): stdout is ONE JSON
    line that fits the driver's tail, with the contract's keys, the roofline object of the dominant kernel, the CPU baseline
    (child process) and the compact summary of the slices a second child process measured; stderr stays short too."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workloads", "6,7"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")
    assert len(r.stdout) + len(r.stderr) < 7000, (len(r.stdout), len(r.stderr))
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "workloads_summary", "library", "distributed"):
        assert key in d, key
    assert "dropped_for_length" not in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert d["config"]["workload"].startswith("BASELINE configs[1]") and d["config"]["global_batch"] == 16
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
    assert ro["achieved"] > 0 and ro["peak"] > 0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    assert "traffic" in ro
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert 0 < d["value"] and abs(d["value"] - 16 / (1000 * d["ms_per_step"] / 1e3)) < 1e-9
    ws = d["workloads_summary"]
    assert [w["id"] for w in ws] == ["c0_f32", "c0_bf16"] and all(w["ms_per_step"] > 0 and "error" not in w for w in ws)
    full = json.load(open(os.path.join(ROOT, d["workloads_file"]["path"])))["workloads"]
    assert len(full) == 2 and full[0]["roofline"]["frac"] > 0 and d["workloads_file"]["child_rc"] == 0


@pytest.mark.gpu
def test_bench_small_variant_line():
    """A non-default shape (no child slices): still one line with roofline + cpu_baseline."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "workloads_summary" not in d and d["roofline"]["achieved"] > 0 and d["cpu_baseline"]["value"] > 0
    assert abs(d["value"] - 2 / (1000 * d["ms_per_step"] / 1e3)) < 1e-9

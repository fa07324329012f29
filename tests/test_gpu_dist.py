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

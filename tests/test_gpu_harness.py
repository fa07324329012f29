"""the degradation + scoring harness (mirror of test_webp_restoration, webp_inference.py:604-797) on the
HIP path: JSON schema, the step rule, and independence of the results from the number of ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# webp_inference.py:771-780 (LPIPS needs pretrained AlexNet weights that cannot be fetched here: those two keys are omitted)
REFERENCE_KEYS = {"compressed_psnr", "compressed_ssim", "compressed_l2", "restored_psnr", "restored_ssim", "restored_l2"}


def _run(world, out_dir):
    env = dict(os.environ, DDPMIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "dist_harness_worker.py")
    if world == 1:
        cmd = [sys.executable, worker, out_dir]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", "29547", worker, out_dir]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("HARNESS-JSON ")]
    assert len(line) == 1
    return json.loads(line[0][len("HARNESS-JSON "):])


@pytest.mark.gpu
def test_restoration_harness_schema_and_rank_independence(tmp_path):
    one = _run(1, str(tmp_path / "w1"))
    assert set(one) == {"10", "70"}
    for q, r in one.items():
        assert REFERENCE_KEYS <= set(r), set(r)
        assert r["count"] == 4
        assert 5.0 < r["compressed_psnr"] < 60.0 and 0.0 < r["compressed_ssim"] <= 1.0 and r["compressed_l2"] > 0
        assert abs(r["psnr_gain"] - (r["restored_psnr"] - r["compressed_psnr"])) < 1e-9
    assert one["70"]["compressed_psnr"] > one["10"]["compressed_psnr"]          # higher quality, less damage
    saved = json.load(open(tmp_path / "w1" / "metrics_summary.json"))            # webp_inference.py:792
    assert saved == one
    # two ranks (3 batches -> 2 + 1): every image keeps its global index, hence its noise stream and its restoration
    two = _run(2, str(tmp_path / "w2"))
    for q in one:
        for k in one[q]:
            assert abs(one[q][k] - two[q][k]) < 1e-9, (q, k, one[q][k], two[q][k])

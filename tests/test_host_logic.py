"""CPU: host-side logic of the package -- state-dict table, synthetic generators, codec module, sharding."""
import pytest
import numpy as np

from ddpm_image_restoration_amd import codec, dist, spec, synth
from oracle import ddpm_oracle as O


def test_state_spec_counts():
    # SURVEY.md / BASELINE.md: 114 398 409 parameters (WebP = JPEG), 158 284 137 (AVIF); 356 / 634 tensors
    assert spec.param_count(spec.ARCH_WEBP) == 114398409
    assert spec.param_count(spec.ARCH_JPEG) == 114398409
    assert spec.param_count(spec.ARCH_AVIF) == 158284137
    assert len(spec.state_spec(spec.ARCH_WEBP)) == 356
    assert len(spec.state_spec(spec.ARCH_AVIF)) == 634
    assert spec.gn_groups(3) == 3 and spec.gn_groups(64) == 8 and spec.gn_groups(7) == 7


def test_synth_is_deterministic_and_named():
    a = synth.synth_tensor("down2.conv1.weight", (128, 64, 3, 3), 0)
    b = synth.synth_tensor("down2.conv1.weight", (128, 64, 3, 3), 0)
    c = synth.synth_tensor("down3.conv1.weight", (128, 64, 3, 3), 0)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    d = synth.dct_matrix(8)
    assert np.allclose(d @ d.T, np.eye(8), atol=1e-6)
    img = synth.synth_images(2, 64)
    assert img.shape == (2, 3, 64, 64) and img.min() >= -1 and img.max() <= 1


def test_codec_module_matches_oracle_bit_exact(golden):
    g = golden("codec")
    for kind in ("webp", "jpeg", "avif"):
        for q in (10, 50):
            u8 = O.to_u8(g["in_32"])
            assert np.array_equal(codec.roundtrip_u8(u8, q, kind), O.codec_u8(u8, q, kind))
            assert np.array_equal(O.from_u8(codec.roundtrip_u8(u8, q, kind)), g[f"{kind}_q{q}_32"])


def test_shard_range_partitions():
    for n in (1, 7, 16, 128, 130):
        for w in (1, 2, 3, 4, 8):
            spans = [dist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_harness_step_rule():
    from ddpm_image_restoration_amd.harness import steps_for_quality
    # webp_inference.py:678-679 clamp(100-q, 20, 80); avif_inference.py:534-535 clamp(100-q, 15, 75)
    assert [steps_for_quality(spec.ARCH_WEBP, q) for q in (0, 10, 50, 90)] == [80, 80, 50, 20]
    assert [steps_for_quality(spec.ARCH_AVIF, q) for q in (1, 30, 90)] == [75, 70, 15]
    # the reference computes int((100 - q) / 100 * 100) in floating point: one step fewer than 100 - q at these qualities
    assert [steps_for_quality(spec.ARCH_WEBP, q) for q in (42, 43, 71)] == [57, 56, 28]
    assert [steps_for_quality(spec.ARCH_AVIF, q) for q in (42, 43, 71)] == [57, 56, 28]
    assert all(steps_for_quality(spec.ARCH_JPEG, q) == max(20, min(80, int((100 - q) / 100 * 100))) for q in range(0, 101))


def test_harness_ssim_matches_independent_restatement():
    """SSIM (pytorch_msssim algorithm, webp_inference.py:698) vs a scipy restatement; identity gives exactly 1."""
    import numpy as np
    import torch
    from scipy.ndimage import correlate1d
    from ddpm_image_restoration_amd.harness import ssim
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (2, 3, 40, 24)).astype(np.float32)
    b = np.clip(a + rng.normal(0, 0.2, a.shape), -1.2, 1.2).astype(np.float32)
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2)); g /= g.sum()

    def blur(z):
        z = correlate1d(z, g, axis=2, mode="constant")[:, :, 5:-5]
        return correlate1d(z, g, axis=3, mode="constant")[:, :, :, 5:-5]
    x = np.clip(a * 0.5 + 0.5, 0, 1).astype(np.float64); y = np.clip(b * 0.5 + 0.5, 0, 1).astype(np.float64)
    mx, my = blur(x), blur(y)
    sx, sy, sxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
    want = (((2 * mx * my + 1e-4) / (mx * mx + my * my + 1e-4)) * ((2 * sxy + 9e-4) / (sx + sy + 9e-4))).mean()
    got = ssim(torch.from_numpy(a), torch.from_numpy(b))
    assert abs(got - want) < 1e-9
    assert ssim(torch.from_numpy(a), torch.from_numpy(a)) == pytest.approx(1.0, abs=1e-12)
    assert abs(ssim(torch.from_numpy(a), torch.from_numpy(b)) - ssim(torch.from_numpy(b), torch.from_numpy(a))) < 1e-12
    # a dim smaller than the window is left unsmoothed (pytorch_msssim behaviour)
    small = torch.from_numpy(a[:, :, :8, :8])
    assert ssim(small, small) == pytest.approx(1.0, abs=1e-12)


def test_avif_worker_process_pool_matches_in_process_codec(monkeypatch):
    """The AVIF degradation operator spread over worker processes (shared-memory batch) returns exactly what the
    in-process Pillow calls of avif_inference.py:64-98 return; DDPMIR_AVIF_WORKERS=0 keeps everything in-process."""
    u8 = np.clip(synth.synth_images(5, 32, seed=11) * 127.5 + 127.5, 0, 255).astype(np.uint8)
    monkeypatch.setenv("DDPMIR_AVIF_WORKERS", "3")
    pooled = codec.roundtrip_u8(u8, 10, "avif")
    again = codec.roundtrip_u8(u8[:4], 40, "avif")          # pool reuse, smaller batch, other quality
    monkeypatch.setenv("DDPMIR_AVIF_WORKERS", "0")
    assert np.array_equal(pooled, codec.roundtrip_u8(u8, 10, "avif"))
    assert np.array_equal(again, codec.roundtrip_u8(u8[:4], 40, "avif"))
    assert np.array_equal(pooled[0], codec._one(u8[0], 10, "avif"))
    codec._shutdown_procs()


def test_bench_reads_roofline_traffic_from_the_committed_profile():
    """bench.py's roofline.traffic is not a literal: it comes from profiles/*traffic*.json (tools/pmc_summary.py --json)
    together with its provenance; an unknown workload has no traffic figure."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(bench)
    val, src = bench.committed_traffic("jpeg_bf16_256_B16_all")
    assert val and val > 1e8 and src["file"].startswith("profiles/") and "bench.py" in src["command"] and "flash_bf16" in src["kernel"]
    assert bench.committed_traffic("jpeg_bf16_64_B2_all") == (None, None)
    # algorithmic bytes of that launch (Q + packed K/V + bf16 output, B=16, 65 536 tokens, 4 heads of 16): the measured
    # traffic must be at least that and, with K/V re-streamed per workgroup generation, a small multiple of it
    algo = 16 * 65536 * 64 * 2 * 2 + 16 * 4 * 2048 * 2112
    assert algo <= val < 6 * algo


def test_packed_blob_header_fields():
    from ddpm_image_restoration_amd import model
    assert model.PACKED_MAGIC == b"DDPMIRW\x00" and len(model.PACKED_MAGIC) == 8


def _build_and_run_host_check(tmp_path, source, marker):
    """tests/c_abi/<source>: host code of csrc/common.h, compiled with hipcc (no GPU needed) and run here."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / source.replace(".cpp", ""))
    r = subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(root, "ddpm_image_restoration_amd", "csrc"),
                        os.path.join(root, "tests", "c_abi", source), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and marker in r.stdout, r.stdout[-500:]


def test_fastdiv_magic_numbers_are_exact(tmp_path):
    """csrc/common.h FastDiv: the magic-number division that turns a pixel index into (sample, row, column) in the GEMM, strip and
    pointwise kernels must equal n / d for every n < 2^31: exhaustive for small n, around every multiple of d, 200 000 random n per
    divisor, 39 divisors incl. every H*W / W / channel-slot count the models use."""
    _build_and_run_host_check(tmp_path, "fastdiv_check.cpp", "FASTDIV-OK")


def test_e4m3_weight_packer_rounds_to_nearest_even(tmp_path):
    """csrc/common.h f32_to_e4m3 (fp8 mode's weight packer, host twin of v_cvt_pk_fp8_f32): equals the nearest OCP e4m3fn code by
    brute force -- every code, every midpoint between neighbours and one ulp either side (ties to even), 400 000 random values
    over 2^-12 .. 2^9, saturation at 448, infinities, NaN."""
    _build_and_run_host_check(tmp_path, "e4m3_check.cpp", "E4M3-OK")


def test_bf16_grade_gelu_formula():
    """gemm16.hip gelu_bf16_grade (the GroupNorm-apply pass of bf16 / fp8 mode): erf by Abramowitz & Stegun 7.1.26 evaluated in
    fp32.  Restated here in numpy fp32: absolute error against exact erf-GELU (webp_inference.py:313, nn.GELU()) below 5e-7 over
    [-12, 12], and on unit-normal inputs (what GroupNorm hands it) fewer than 1 in 5000 bf16-rounded results differ from the
    bf16-rounded exact value -- three orders below bf16's own rounding error."""
    from scipy.special import erf
    f = np.float32

    def gelu16(v):
        x = np.abs(v) * f(0.70710678118654752440)
        t = (f(1) / (f(0.3275911) * x + f(1))).astype(np.float32)
        poly = (t * (t * (t * (t * (t * f(1.061405429) + f(-1.453152027)) + f(1.421413741)) + f(-0.284496736)) + f(0.254829592))).astype(np.float32)
        e = (f(1) - poly * np.exp(-x * x).astype(np.float32)).astype(np.float32)
        return (f(0.5) * v * (f(1) + np.copysign(e, v))).astype(np.float32)

    def exact(v):
        return 0.5 * v.astype(np.float64) * (1 + erf(v.astype(np.float64) / np.sqrt(2)))

    def bf16(a):
        u = np.ascontiguousarray(a, np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    v = np.linspace(-12, 12, 400001).astype(np.float32)
    assert np.abs(gelu16(v) - exact(v)).max() < 5e-7
    v = np.random.default_rng(0).standard_normal(1000000).astype(np.float32)
    a, b = bf16(gelu16(v)), bf16(exact(v).astype(np.float32))
    assert (a != b).mean() < 2e-4 and np.abs(a - b).mean() < 1e-3 * np.abs(b - exact(v)).mean()

"""Whole-path parity on the GPU through the C ABI: U-Net forward and DDRM loop against the golden vectors
produced by the reference (tests/golden) and against the numpy oracle at sizes the goldens do not hold."""
import numpy as np
import pytest

from ddpm_image_restoration_amd import spec, synth
from oracle import ddpm_oracle as O

pytestmark = pytest.mark.gpu

ARCHS = [("webp", spec.ARCH_WEBP), ("jpeg", spec.ARCH_JPEG), ("avif", spec.ARCH_AVIF)]
TOL = 1e-3   # north_star: max-abs pixel error within 1e-3 in fp32


@pytest.fixture(scope="module")
def models(synth_state):
    import torch
    from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
    cache = {}

    def get(arch, **kw):
        key = (arch, tuple(sorted(kw.items())))
        if key not in cache:
            m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), **kw)
            m.load_state_dict(synth_state(arch))
            cache[key] = m
        return cache[key]

    return get


def tt(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name,arch", ARCHS)
def test_forward_golden_32(name, arch, golden, models):
    g = golden(name)
    m = models(arch)
    y = m(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"])).cpu().numpy()
    err = np.abs(y - g["fwd32_y"]).max()
    assert err < TOL, err
    assert m.param_count() == spec.param_count(arch)


@pytest.mark.parametrize("name,arch", ARCHS)
def test_forward_golden_64_and_nonsquare(name, arch, golden, models):
    g = golden(name)
    m = models(arch)
    y = m(tt(g["fwd64_x"]), tt(g["fwd64_t"]), tt(g["fwd64_t"])).cpu().numpy()
    assert np.abs(y - g["fwd64_y"]).max() < TOL
    y = m(tt(g["fwd96_x"]), tt(g["fwd64_t"]), None).cpu().numpy()       # compression_level=None -> t
    assert np.abs(y - g["fwd96_y"]).max() < TOL


@pytest.mark.parametrize("name,arch", ARCHS)
def test_forward_second_weight_seed(name, arch, golden, synth_state):
    """An independent set of reference outputs (weights of synth seed 1, other images, t near 0 and 1, compression_level
    != t, a 64 x 32 image; tests/golden/make_golden_seed1.py): fp32 within 1e-3, bf16 mode within its budget."""
    import torch
    from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
    g = golden("seed1_fwd")
    for dtype, tol_max, tol_mean in (("f32", TOL, TOL), ("bf16", 2e-2, 2e-3)):
        m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), dtype=dtype)
        m.load_state_dict(synth_state(arch, 1))
        for xs, ts, cls, ys in ((name + "_x", name + "_t", name + "_cl", name + "_y"), (name + "_x2", name + "_t2", name + "_t2", name + "_y2")):
            e = np.abs(m(tt(g[xs]), tt(g[ts]), tt(g[cls])).cpu().numpy() - g[ys])
            assert e.max() < tol_max and e.mean() < tol_mean, (dtype, xs, e.max(), e.mean())
        del m


@pytest.mark.parametrize("name,arch", ARCHS)
def test_forward_bf16_mode(name, arch, golden, models):
    """dtype='bf16': matrix-core contractions on bf16 operands, fp32 accumulation / statistics / storage.
    Tolerance per SURVEY.md H2: teacher-forced x_theta within 2e-2 of the fp32 reference."""
    g = golden(name)
    m = models(arch, dtype="bf16")
    y = m(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"])).cpu().numpy()
    err = np.abs(y - g["fwd32_y"])
    assert err.max() < 2e-2 and err.mean() < 2e-3, (err.max(), err.mean())
    y = m(tt(g["fwd64_x"]), tt(g["fwd64_t"]), tt(g["fwd64_t"])).cpu().numpy()
    err = np.abs(y - g["fwd64_y"])
    assert err.max() < 2e-2 and err.mean() < 2e-3, (err.max(), err.mean())
    # non-square 96x64: level maps 96x64 ... 3x2 (ragged GEMM tiles, maps smaller than a DCT block, 6-token attention)
    y = m(tt(g["fwd96_x"]), tt(g["fwd64_t"]), None).cpu().numpy()
    err = np.abs(y - g["fwd96_y"])
    assert err.max() < 2e-2 and err.mean() < 2e-3, (err.max(), err.mean())


@pytest.mark.parametrize("name,arch", ARCHS)
def test_forward_graph_replay_matches_eager(name, arch, golden, models):
    g = golden(name)
    y0 = models(arch)(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"])).cpu().numpy()
    mg = models(arch, use_graph=True)
    for _ in range(3):   # capture, then two replays
        y1 = mg(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"])).cpu().numpy()
    assert np.array_equal(y0, y1)


@pytest.mark.parametrize("name,arch", ARCHS)
def test_sampler_teacher_forced_golden(name, arch, golden, models):
    g = golden(name)
    m = models(arch)
    steps, b = int(g["samp_steps"]), g["samp_y"].shape[0]
    for i in g["samp_keep"]:
        t = (np.full((b,), int(i), np.float32) / np.float32(steps)).astype(np.float32)
        th = m(tt(g[f"samp_xt_in_{int(i)}"]), tt(t), tt(t)).cpu().numpy()
        assert np.abs(th - g[f"samp_theta_{int(i)}"]).max() < TOL, int(i)


@pytest.mark.parametrize("name,arch", ARCHS)
def test_sampler_teacher_forced_bf16(name, arch, golden, models):
    """The benchmarked arithmetic (bf16 matrix-core operands) along the reference trajectories: fed the reference's own x_t
    at every kept step, x_theta stays within SURVEY.md H2's teacher-forced budget (2e-2 max, 2e-3 mean; measured
    0.8-1.4e-2 max, 1.5-1.7e-3 mean for all three families, tools/parity_r2.py)."""
    g = golden(name)
    m = models(arch, dtype="bf16")
    steps, b = int(g["samp_steps"]), g["samp_y"].shape[0]
    for i in g["samp_keep"]:
        t = (np.full((b,), int(i), np.float32) / np.float32(steps)).astype(np.float32)
        th = m(tt(g[f"samp_xt_in_{int(i)}"]), tt(t), tt(t)).cpu().numpy()
        e = np.abs(th - g[f"samp_theta_{int(i)}"])
        assert e.max() < 2e-2 and e.mean() < 2e-3, (int(i), e.max(), e.mean())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_timestep_graph_replay_matches_eager_device_codec(dtype, models):
    """north_star: the T-step loop as a captured hipGraph.  use_graph=True replays {t, U-Net forward, uint8 quantise, device
    JPEG, Philox noise, DDRM update (+ phase fix), step advance} with the per-step scalars in a device-side parameter block;
    the eager loop launches the same kernels with the same arguments, so the results are bit-identical -- with device noise,
    with injected noise, across the phase-fix steps (q10: every 5th index) and through the final no-noise step."""
    import torch
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler
    y = tt(O.compress(synth.synth_images(3, 32, seed=81), 10, "jpeg"))
    se, sg = DDRMJPEGSampler(models(spec.ARCH_JPEG, dtype=dtype)), DDRMJPEGSampler(models(spec.ARCH_JPEG, dtype=dtype, use_graph=True))
    a, ta = se.sample(y, 10, steps=12, seed=9, image_offset=5, trace=True)
    b, tb = sg.sample(y, 10, steps=12, seed=9, image_offset=5, trace=True)
    assert torch.equal(a, b) and torch.equal(ta, tb)
    assert torch.equal(sg.sample(y, 10, steps=12, seed=9, image_offset=5), a)          # replay of the cached graphs
    assert not torch.equal(sg.sample(y, 10, steps=12, seed=10, image_offset=5), a)     # the seed lives in the parameter block
    noise = tt(synth.synth_noise(7, 3, 32, 32, seed=12))
    assert torch.equal(se.sample(y, 10, steps=7, noise=noise), sg.sample(y, 10, steps=7, noise=noise))
    # other baked-in scalars (quality -> quantisation tables, no phase fix at q50) re-capture
    y50 = tt(O.compress(synth.synth_images(3, 32, seed=81), 50, "jpeg"))
    assert torch.equal(se.sample(y50, 50, steps=6, seed=2), sg.sample(y50, 50, steps=6, seed=2))
    # a slice of a longer schedule, as bench.py times it
    kw = dict(steps=1000, start_step=996, num_steps=4, seed=4)
    assert torch.equal(se.sample(y, 10, **kw), sg.sample(y, 10, **kw))


def test_timestep_graph_replay_matches_eager_host_codec(models):
    """Host-codec families: each of the two software-pipelined micro-batch lanes replays its own pair of captured
    half-steps around the codec callback."""
    import torch
    from ddpm_image_restoration_amd.sampler import DDRMWebPSampler
    y = tt(O.compress(synth.synth_images(4, 32, seed=82), 10, "webp"))
    a = DDRMWebPSampler(models(spec.ARCH_WEBP)).sample(y, 10, steps=11, seed=3)
    b = DDRMWebPSampler(models(spec.ARCH_WEBP, use_graph=True)).sample(y, 10, steps=11, seed=3)
    assert torch.equal(a, b)
    c = DDRMWebPSampler(models(spec.ARCH_WEBP, use_graph=True)).sample(y[:3], 10, steps=11, seed=3)   # odd batch: one lane
    assert torch.equal(c, a[:3])


def test_two_contexts_in_one_process_run_concurrently(synth_state):
    """include/ddpmir.h: "a context is bound to one GPU and is not thread-safe; different contexts may run concurrently".
    Two contexts (own streams, own workspaces) driven from two host threads at the same time give what they give alone."""
    import threading
    import torch
    from ddpm_image_restoration_amd.model import JPEGDiffusionModel, WebPDiffusionModel
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler, DDRMWebPSampler
    dev = torch.device("cuda", 0)
    ma = JPEGDiffusionModel(device=dev, dtype="bf16"); ma.load_state_dict(synth_state(spec.ARCH_JPEG))
    mb = WebPDiffusionModel(device=dev); mb.load_state_dict(synth_state(spec.ARCH_WEBP))
    ya = tt(O.compress(synth.synth_images(2, 64, seed=91), 10, "jpeg"))
    yb = tt(O.compress(synth.synth_images(2, 32, seed=92), 10, "webp"))
    sa, sb = DDRMJPEGSampler(ma), DDRMWebPSampler(mb)
    ra, rb = sa.sample(ya, 10, steps=10, seed=1), sb.sample(yb, 10, steps=10, seed=2)
    got, errs = {}, []

    def run(key, s, y, seed):
        try:
            for _ in range(3):
                got[key] = s.sample(y, 10, steps=10, seed=seed)
        except Exception as e:   # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=run, args=("a", sa, ya, 1)), threading.Thread(target=run, args=("b", sb, yb, 2))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert torch.equal(got["a"], ra) and torch.equal(got["b"], rb)


@pytest.mark.parametrize("name,arch", ARCHS)
def test_sampler_end_to_end_golden(name, arch, golden, models):
    """Full trajectories with injected noise and the host codec callback against the reference's: EVERY image stays within
    1e-3 of the reference at every kept step and at the end (measured 2-5e-6 for all three families), restored PSNR equal to
    2 decimals.
    The loop is chaotic through `.to(uint8)` truncation + lossy codec, but an fp32 rounding-level difference
    in x_theta only flips single uint8 pixels (about one per step and batch here), which the q10 codecs absorb.  What did
    throw one JPEG image off in round 1 was the phase fix: fft2(y) of that image has a bin that is zero up to 6e-8, and the
    phase of such a bin is decided by the transform's rounding; arg(fft2(y)) now comes from a double-precision transform
    (test_phase_fix_of_a_numerically_zero_bin)."""
    from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
    g = golden(name)
    m = models(arch)
    steps, q, b = int(g["samp_steps"]), int(g["samp_q"]), g["samp_y"].shape[0]
    noise = synth.synth_noise(steps, b, 32, 32, seed=7)
    out, tr = SAMPLER_BY_ARCH[arch](m).sample(tt(g["samp_y"]), q, steps=steps, noise=tt(noise), trace=True,
                                              device_codec=False)
    out, tr = out.cpu().numpy(), tr.cpu().numpy()
    for i in (int(v) for v in g["samp_keep"]):
        assert np.abs(tr[steps - 1 - i, 0] - g[f"samp_theta_{i}"]).max() < TOL, i
        if i < steps - 1:
            assert np.abs(tr[steps - 2 - i, 1] - g[f"samp_xt_in_{i}"]).max() < TOL, i
    x0 = g["samp_x0"]
    per_img = np.abs(out - g["samp_final"]).reshape(b, -1).max(axis=1)
    assert (per_img < TOL).all(), per_img
    assert round(O.psnr(out, x0), 2) == round(O.psnr(g["samp_final"], x0), 2)
    # q50 run: no phase fix, JPEG 4:4:4
    steps2, q2 = int(g["samp2_steps"]), int(g["samp2_q"])
    n2 = synth.synth_noise(steps2, 2, 32, 32, seed=8)
    out2 = SAMPLER_BY_ARCH[arch](m).sample(tt(g["samp2_y"]), q2, steps=steps2, noise=tt(n2), device_codec=False).cpu().numpy()
    per2 = np.abs(out2 - g["samp2_final"]).reshape(2, -1).max(axis=1)
    assert (per2 < TOL).all(), per2
    assert round(O.psnr(out2, x0[:2]), 2) == round(O.psnr(g["samp2_final"], x0[:2]), 2)


@pytest.mark.parametrize("R,n_batches,bs", [(32, 2, 256), (64, 1, 128)])
def test_restored_psnr_bf16_mode_equals_fp32_mode(R, n_batches, bs, models):
    """north_star's "identical PSNR to 2 decimals" for the arithmetic the benchmark runs: JPEG q10, the harness' full step
    schedule (80 steps), device noise, bf16 mode against fp32 parity mode on the same degraded images.  The loop is chaotic
    (H2), so single images differ -- measured over 1024 images at 32x32: per-image PSNR difference mean +0.010 dB, std 0.13,
    max 0.52; over 256 images at 64x64: mean -0.001, std 0.06, max 0.20 (tools/parity_r2.py psnrN) -- a zero-mean scatter:
    the mean restored PSNR of the two modes agrees to within its standard error, i.e. to 2 decimals over a test set."""
    from ddpm_image_restoration_amd import harness
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler
    import torch
    s32, s16 = DDRMJPEGSampler(models(spec.ARCH_JPEG)), DDRMJPEGSampler(models(spec.ARCH_JPEG, dtype="bf16"))
    steps = harness.steps_for_quality(spec.ARCH_JPEG, 10)
    d = []
    for k in range(n_batches):
        x0 = synth.synth_images(bs, R, seed=900 + 10 * R + k)
        y = harness.COMPRESS[spec.ARCH_JPEG](torch.from_numpy(x0), 10).cuda()
        r32 = s32.sample(y, 10, steps=steps, seed=3, image_offset=k * bs).cpu().numpy()
        r16 = s16.sample(y, 10, steps=steps, seed=3, image_offset=k * bs).cpu().numpy()

        def psnr_each(a):
            a01, b01 = np.clip(a * 0.5 + 0.5, 0, 1), np.clip(x0 * 0.5 + 0.5, 0, 1)
            return -10 * np.log10(((a01 - b01) ** 2).reshape(bs, -1).mean(axis=1) + 1e-8)
        d.append(psnr_each(r16) - psnr_each(r32))
    d = np.concatenate(d)
    se = d.std() / np.sqrt(d.size)
    print(f"{R}x{R}, {d.size} images: bf16 - fp32 restored PSNR mean {d.mean():+.4f} dB (standard error {se:.4f}), max |d| {np.abs(d).max():.3f}")
    assert abs(d.mean()) < 0.02, d.mean()            # 3.5 standard errors at 32x32 (N = 512), 3.7 at 64x64 (N = 128)
    assert np.abs(d).max() < 1.0                     # no image falls out: the scatter is that of the chaotic loop
    assert d.std() < 0.25


def test_sampler_jpeg_every_step_49_to_39(golden, models):
    """tests/golden/jpeg_steps49_39.npz holds x_t and x_theta of the reference for EVERY step i = 49 ... 39 of the config-1
    run (the stretch with two phase fixes, i = 45 and 40, where round 1 lost image 1): all four images track."""
    import os
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler
    g = golden("jpeg")
    gs = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_steps49_39.npz"))
    steps, q, b = int(g["samp_steps"]), int(g["samp_q"]), g["samp_y"].shape[0]
    noise = synth.synth_noise(steps, b, 32, 32, seed=7)
    _, tr = DDRMJPEGSampler(models(spec.ARCH_JPEG)).sample(tt(g["samp_y"]), q, steps=steps, noise=tt(noise), trace=True,
                                                          num_steps=11)
    tr = tr.cpu().numpy()
    for i in range(49, 38, -1):
        assert np.abs(tr[49 - i, 0] - gs[f"theta_{i}"]).max() < TOL, i
        if i < 49:
            assert np.abs(tr[48 - i, 1] - gs[f"xt_in_{i}"]).max() < TOL, i


def test_phase_fix_of_a_numerically_zero_bin(golden, models):
    """phase_consistency (webp_inference.py:531-550) takes arg(fft2(ref)).  Image 1 of the JPEG golden batch has a spectrum
    bin of magnitude 6e-8 (exactly representable inputs that cancel): the reference's FFT returns +1.2e-7 there, phase 0.
    A transform with fp32 direct-summation error (1e-5) gets the sign of that bin by chance and moves every pixel by
    2 * 0.3 * |X_bin| / (H W) ~ 1e-3.  The library computes this one transform in double, so the result matches the
    reference's own phase_consistency output for that image to fp32 accuracy."""
    from ddpm_image_restoration_amd.sampler import phase_consistency
    g = golden("jpeg")
    y = g["samp_y"]
    mags = np.abs(np.fft.fft2(y.astype(np.float64))).reshape(y.shape[0], -1).min(axis=1)
    assert mags[1] < 1e-6 and (np.delete(mags, 1) > 1e-3).all()          # the fixture really has the degenerate bin
    x = g["samp_xt_in_40"]                                                # a noisy x_t of the same run
    # the oracle's phase_consistency follows the reference (numpy's pocketfft == torch's CPU FFT kernel) -- checked against
    # the reference's own pc_out in tests/test_oracle_golden.py
    want = O.phase_consistency(x, y, 0.7)
    got = phase_consistency(models(spec.ARCH_JPEG), tt(x), tt(y), alpha=0.7).cpu().numpy()
    assert np.abs(got - want).reshape(y.shape[0], -1).max(axis=1).max() < 2e-5


@pytest.mark.parametrize("name,arch", ARCHS)
def test_phase_consistency_golden(name, arch, golden, models):
    from ddpm_image_restoration_amd.sampler import phase_consistency
    g = golden(name)
    alpha = O.SAMPLER[arch][3]
    out = phase_consistency(models(arch), tt(g["pc_x"]), tt(g["pc_ref"]), alpha).cpu().numpy()
    assert np.abs(out - g["pc_out"]).max() < 1e-5


def test_phase_consistency_wide_image_past_the_default_lds_limit(models):
    """The fp64 reference-phase kernels stage (4 + 2) * W doubles per workgroup: past W = 1365 that exceeds the default 64 KiB
    of dynamic LDS and the launch limit is raised per call (up to 160 KiB: W <= 3413) instead of failing."""
    from ddpm_image_restoration_amd.sampler import phase_consistency
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (1, 3, 32, 1408)).astype(np.float32)
    ref = rng.uniform(-1, 1, (1, 3, 32, 1408)).astype(np.float32)
    got = phase_consistency(models(spec.ARCH_JPEG), tt(x), tt(ref), alpha=0.7).cpu().numpy()
    want = O.phase_consistency(x, ref, 0.7)
    assert np.abs(got - want).max() < 2e-4


def test_forward_vs_oracle_128_bottleneck_attention(models, synth_state):
    """attn_levels='bottleneck' variant (BASELINE configs 2-5) at 128x128 against the oracle with the same switch."""
    x = synth.synth_images(1, 128, seed=77)
    t = np.array([0.25], np.float32)
    ref = O.Model(spec.ARCH_JPEG, synth_state(spec.ARCH_JPEG), attn_min_level=5)(x, t, t)
    y = models(spec.ARCH_JPEG, attn_levels="bottleneck")(tt(x), tt(t), tt(t)).cpu().numpy()
    assert np.abs(y - ref).max() < TOL


def test_sampler_philox_shard_independent(models):
    """Device RNG: sampling images [0:4) as one batch or as two shards of 2 gives identical results
    (noise keyed by global image index) -- the property the multi-GPU split relies on."""
    from ddpm_image_restoration_amd.sampler import DDRMWebPSampler
    m = models(spec.ARCH_WEBP)
    y = tt(synth.synth_images(4, 32, seed=5))
    s = DDRMWebPSampler(m)
    full = s.sample(y, 50, steps=4, seed=11).cpu().numpy()
    a = s.sample(y[:2], 50, steps=4, seed=11, image_offset=0).cpu().numpy()
    b = s.sample(y[2:], 50, steps=4, seed=11, image_offset=2).cpu().numpy()
    assert np.array_equal(full, np.concatenate([a, b]))


def test_error_paths(models):
    import torch
    from ddpm_image_restoration_amd.model import WebPDiffusionModel
    m = WebPDiffusionModel(device=torch.device("cuda", 0))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32).cuda(), torch.zeros(1).cuda())        # weights not loaded
    with pytest.raises(RuntimeError):
        m.load_state_dict({"bogus": np.zeros(3, np.float32)})
    with pytest.raises(RuntimeError):
        models(spec.ARCH_WEBP)(torch.zeros(1, 3, 40, 40).cuda(), torch.zeros(1).cuda())   # not a multiple of 32


@pytest.mark.parametrize("quality", [1, 5, 10, 20, 30, 31, 50, 75, 90, 100])
def test_device_jpeg_bit_exact_vs_pillow(quality, models):
    """8(f)1: the on-device JPEG round trip must equal Pillow/libjpeg-turbo bit for bit (both subsamplings)."""
    import torch
    from ddpm_image_restoration_amd.sampler import jpeg_roundtrip_device
    m = models(spec.ARCH_JPEG)
    rng = np.random.default_rng(quality)
    imgs = [O.to_u8(synth.synth_images(2, 64, seed=quality)),                   # smooth field
            rng.integers(0, 256, (2, 3, 32, 96), dtype=np.uint8),               # white noise, non-square
            O.to_u8(synth.synth_images(1, 256, seed=99)),                       # bench size
            np.zeros((1, 3, 32, 32), np.uint8), np.full((1, 3, 32, 32), 255, np.uint8)]
    sat = rng.integers(0, 2, (1, 3, 64, 64)).astype(np.uint8) * 255            # saturated checker noise
    for u8 in imgs + [sat]:
        got = jpeg_roundtrip_device(m, torch.from_numpy(u8), quality).cpu().numpy()
        ref = O.codec_u8(u8, quality, "jpeg")
        assert np.array_equal(got, ref), (quality, u8.shape, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_sampler_device_codec_equals_host_codec(models, golden):
    """The DDRM loop with the device JPEG codec produces exactly what the Pillow callback path produces."""
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler
    g = golden("jpeg")
    m = models(spec.ARCH_JPEG)
    noise = tt(synth.synth_noise(6, 2, 32, 32, seed=8))
    s = DDRMJPEGSampler(m)
    for q in (10, 50):
        y = tt(O.compress(g["samp_x0"][:2], q, "jpeg"))
        a = s.sample(y, q, steps=6, noise=noise, device_codec=False).cpu().numpy()
        b = s.sample(y, q, steps=6, noise=noise, device_codec=True).cpu().numpy()
        assert np.array_equal(a, b), q


def test_eps_unet_predict_noise_and_p_sample_loop(golden):
    """a14: eps-predict UNetDenoiser + ddpm_inference against the reference's own outputs."""
    import torch
    from ddpm_image_restoration_amd.model import UNetDenoiser, ddpm_inference
    g = golden("ddpm_eps")
    net = UNetDenoiser(device=torch.device("cuda", 0))
    net.load_state_dict(synth.synth_eps_unet_state(3))
    eps = net(tt(g["x0"]), torch.tensor([[5.0]])).cpu().numpy()
    assert np.abs(eps - g["eps_t5"]).max() < 1e-4
    out = ddpm_inference(net, tt(g["x0"]), T=8, betas=g["betas"]).cpu().numpy()
    assert np.abs(out - g["out_T8"]).max() < 1e-3
    # batch > 1 and a second size against the oracle
    x = (synth.synth_images(3, 64, seed=2) * 0.5 + 0.5).astype(np.float32)
    ref = O.EpsUnet(synth.synth_eps_unet_state(3))(x, np.array([17.0], np.float32))
    got = net(tt(x), 17.0).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4


def test_weight_blob_adoption_matches_direct_load(golden, synth_state):
    """Multi-GPU start-up path: rank 0 packs the weights into a blob, the other ranks adopt a broadcast copy
    (finalize(blob, fill=False)) without ever seeing the state dict.  Same outputs, bit for bit."""
    import torch
    from ddpm_image_restoration_amd.model import JPEGDiffusionModel
    g = golden("jpeg")
    dev = torch.device("cuda", 0)
    for dtype in ("f32", "bf16"):
        root = JPEGDiffusionModel(device=dev, dtype=dtype)
        blob = torch.empty(root.weight_blob_bytes(), dtype=torch.uint8, device=dev)
        for k, v in synth_state(spec.ARCH_JPEG).items():
            root._load_one(k, v)
        root.finalize(blob=blob, fill=True)
        other = JPEGDiffusionModel(device=dev, dtype=dtype)
        blob2 = blob.clone()                      # stands in for dist.broadcast(blob, src=0)
        other.finalize(blob=blob2, fill=False)
        a = root(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"]))
        b = other(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"]))
        assert torch.equal(a, b)        # both modes are bit-reproducible (split-K partials are folded in a fixed order)
        assert torch.equal(a, root(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"])))
        assert other.param_count() == spec.param_count(spec.ARCH_JPEG)


@pytest.mark.parametrize("name,arch", [("webp", spec.ARCH_WEBP), ("avif", spec.ARCH_AVIF)])
def test_builtin_host_codec_pool_equals_pillow_callback_in_the_loop(name, arch, models):
    """codec == NULL for WebP / AVIF: the library's libwebp / libavif pool inside the two-lane loop gives exactly what the
    callback into Pillow (the reference's own calls) gives -- with the phase fix (q10) and without (q50)."""
    import torch
    from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
    s = SAMPLER_BY_ARCH[arch](models(arch))
    for q, size in ((10, 32), (50, 64)):
        y = tt(O.compress(synth.synth_images(4, size, seed=60 + q), q, name))
        a = s.sample(y, q, steps=6, seed=3, host_codec="pillow")
        b = s.sample(y, q, steps=6, seed=3, host_codec="native")
        assert torch.equal(a, b), (name, q)


def test_bf16_activation_stream_in_the_two_lane_loop(models):
    """fp8 mode's default and bf16 mode's option: the tensors between residual blocks are bf16.  Inside the two-lane host-codec
    loop (WebP: down1 / up5 run per lane on row ranges of the concat buffers, the levels below once for both lanes) the row
    offsets are taken in bf16 elements: x_theta of a timestep of the T=50 schedule stays within the mode's fence of fp32
    mode's x_theta on the same x_t (a wrong offset would put another image's rows under a lane)."""
    import torch
    from ddpm_image_restoration_amd.sampler import DDRMWebPSampler
    y = tt(O.compress(synth.synth_images(4, 64, seed=71), 10, "webp"))
    kw = dict(steps=50, start_step=40, num_steps=1, seed=5, trace=True)
    _, t32 = DDRMWebPSampler(models(spec.ARCH_WEBP)).sample(y, 10, **kw)
    for dtype in ("bf16", "fp8"):
        m = models(spec.ARCH_WEBP, dtype=dtype)
        if dtype == "bf16":
            m.set_option("bf16_activation_stream", 1)
        _, t16 = DDRMWebPSampler(m).sample(y, 10, **kw)
        assert torch.isfinite(t16).all()
        e = (t16[0, 0] - t32[0, 0]).abs()
        lim = (2.5e-2, 2.5e-3) if dtype == "bf16" else (0.17, 1.1e-2)
        assert float(e.max()) < lim[0] and float(e.mean()) < lim[1], (dtype, float(e.max()), float(e.mean()))
        if dtype == "bf16":
            m.set_option("bf16_activation_stream", -1)


def test_avif_per_image_jpeg_fallback_in_the_builtin_pool(models, monkeypatch):
    """avif_inference.py:88-95: an image whose AVIF round trip raises is silently round-tripped as JPEG instead (Pillow's plain
    save(format="JPEG", quality=q): 4:2:0 at every quality).  The built-in libavif pool keeps that behaviour per image: the
    failed image's decoded pixels come from the device JPEG simulator, the rest from libavif.  $DDPMIR_TEST_CODEC_FAIL makes
    image 1 of every lane batch fail; the expected result is the callback path with Pillow doing exactly what the reference
    would have done for those images.  q50: the JPEG FAMILY's rule would pick 4:4:4 there -- the fallback must not."""
    import io
    import torch
    from PIL import Image
    from ddpm_image_restoration_amd import codec
    from ddpm_image_restoration_amd.sampler import DDRMAVIFSampler
    s = DDRMAVIFSampler(models(spec.ARCH_AVIF))

    def reference_with_failure(src, quality, kind, out=None):
        assert kind == "avif"
        res = out if out is not None else np.empty_like(src)
        for j in range(src.shape[0]):
            if j == 1:   # "the AVIF encoder raised" -> the reference's except branch
                buf = io.BytesIO()
                Image.fromarray(np.ascontiguousarray(src[j].transpose(1, 2, 0))).save(buf, format="JPEG", quality=max(1, min(100, int(quality))))
                buf.seek(0)
                res[j] = np.asarray(Image.open(buf).convert("RGB"), dtype=np.uint8).transpose(2, 0, 1)
            else:
                res[j] = codec._one(src[j], quality, "avif")
        return res

    for q, size in ((50, 64), (10, 32)):
        y = tt(O.compress(synth.synth_images(4, size, seed=80 + q), q, "avif"))
        monkeypatch.setattr(codec, "roundtrip_u8", reference_with_failure)
        want = s.sample(y, q, steps=4, seed=3, host_codec="pillow")
        monkeypatch.undo()
        clean = s.sample(y, q, steps=4, seed=3, host_codec="native")
        monkeypatch.setenv("DDPMIR_TEST_CODEC_FAIL", "1")
        got = s.sample(y, q, steps=4, seed=3, host_codec="native")
        monkeypatch.delenv("DDPMIR_TEST_CODEC_FAIL")
        assert torch.equal(got, want), (q, float((got - want).abs().max()))
        assert not torch.equal(got[1], clean[1]) and torch.equal(got[0], clean[0])   # lanes of 2: images 1 and 3 took the fallback


def test_half_precision_checkpoints_load_exactly(golden, synth_state):
    """ddpmir_load_tensor(dtype): a checkpoint stored in bf16 / fp16 crosses the C ABI in its storage type and is widened
    exactly -- same outputs as loading the float32 tensors holding the same (rounded) values."""
    import torch
    from ddpm_image_restoration_amd.model import WebPDiffusionModel
    g = golden("webp")
    dev = torch.device("cuda", 0)
    for half in (torch.bfloat16, torch.float16):
        sd_h = {k: torch.from_numpy(v).to(half) for k, v in synth_state(spec.ARCH_WEBP).items()}
        a = WebPDiffusionModel(device=dev); a.load_state_dict(sd_h)
        b = WebPDiffusionModel(device=dev); b.load_state_dict({k: v.float() for k, v in sd_h.items()})
        x, t, cl = tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"])
        assert torch.equal(a(x, t, cl), b(x, t, cl)), half


def test_model_object_guards(tmp_path, synth_state):
    """finalize twice, a packed blob of another configuration, and .to() another GPU index are errors, not garbage."""
    import torch
    from ddpm_image_restoration_amd.model import JPEGDiffusionModel, WebPDiffusionModel
    dev = torch.device("cuda", 0)
    m = JPEGDiffusionModel(device=dev, dtype="bf16")
    blob = torch.empty(m.weight_blob_bytes(), dtype=torch.uint8, device=dev)
    for k, v in synth_state(spec.ARCH_JPEG).items():
        m._load_one(k, v)
    m.finalize(blob=blob, fill=True)
    with pytest.raises(RuntimeError, match="already finalized"):
        m.finalize(blob=blob, fill=False)
    m.save_packed(str(tmp_path / "jpeg_bf16.npy"))
    JPEGDiffusionModel(device=dev, dtype="bf16").load_packed(str(tmp_path / "jpeg_bf16.npy"))        # same configuration: fine
    for other in (JPEGDiffusionModel(device=dev, dtype="f32"), JPEGDiffusionModel(device=dev, dtype="bf16", attn_levels="bottleneck")):
        with pytest.raises(RuntimeError, match="does not match"):
            other.load_packed(str(tmp_path / "jpeg_bf16.npy"))
    with pytest.raises(RuntimeError, match="does not match"):        # WebP and JPEG blobs have the same byte count
        WebPDiffusionModel(device=dev, dtype="bf16").load_packed(str(tmp_path / "jpeg_bf16.npy"))
    assert m.to("cuda") is m and m.to(torch.device("cuda", 0)) is m
    with pytest.raises(RuntimeError, match="lives on"):
        m.to("cuda:1")
    with pytest.raises(RuntimeError):
        m.to("cpu")


def test_checkpoint_converter_roundtrip(tmp_path, golden, synth_state):
    """8(f)4: .pth state dict -> packed blob file -> model; wrapped and bare state dicts; name validation."""
    import torch
    from ddpm_image_restoration_amd.model import WebPDiffusionModel, convert_checkpoint, load_checkpoint
    g = golden("webp")
    dev = torch.device("cuda", 0)
    sd = {k: torch.from_numpy(v) for k, v in synth_state(spec.ARCH_WEBP).items()}
    torch.save({"epoch": 3, "model_state_dict": sd}, tmp_path / "wrapped.pth")
    torch.save(sd, tmp_path / "bare.pth")
    a = load_checkpoint(WebPDiffusionModel(device=dev), str(tmp_path / "wrapped.pth"))
    convert_checkpoint(WebPDiffusionModel(device=dev), str(tmp_path / "bare.pth"), str(tmp_path / "packed.npy"))
    b = WebPDiffusionModel(device=dev).load_packed(str(tmp_path / "packed.npy"))
    ya = a(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"]))
    yb = b(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"]))
    assert torch.equal(ya, yb)
    assert np.abs(ya.cpu().numpy() - g["fwd32_y"]).max() < TOL
    bad = dict(sd); bad.pop("down3.conv2.bias")
    torch.save(bad, tmp_path / "bad.pth")
    with pytest.raises(RuntimeError, match="missing"):
        convert_checkpoint(WebPDiffusionModel(device=dev), str(tmp_path / "bad.pth"), str(tmp_path / "x.npy"))


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json's full size (256x256, attention over all 65 536 tokens): the numpy oracle needs ~13 s per image and step
# here, so parity at this size is checked through properties that do not depend on the size.
def test_full_size_forward_properties(models):
    import torch
    m32, m16 = models(spec.ARCH_JPEG), models(spec.ARCH_JPEG, dtype="bf16")
    x = tt(synth.synth_images(2, 256, seed=21))
    t = torch.tensor([0.31, 0.87], device="cuda")
    y32 = m32(x, t, t)
    assert torch.isfinite(y32).all() and float(y32.abs().max()) <= 1.0           # tanh head
    # images of a batch do not interact: each one alone gives bit-identical results (fp32 mode is deterministic)
    assert torch.equal(m32(x[:1], t[:1], t[:1]), y32[:1])
    assert torch.equal(m32(x[1:], t[1:], t[1:]), y32[1:])
    # production mode (bf16 operands, fused qkv packing, split-K) against parity mode at full size, SURVEY H2 tolerance
    y16 = m16(x, t, t)
    err = (y16 - y32).abs()
    assert float(err.max()) < 2e-2 and float(err.mean()) < 2e-3, (float(err.max()), float(err.mean()))


def test_full_size_attention_is_a_convex_combination():
    """N = 65 536 keys, head_dim 16: softmax rows sum to one, so a value tensor that is constant per channel must come
    back unchanged whatever q and k are -- exercises the deferred rescale and the ones-row denominator over 2048 tiles."""
    from ddpm_image_restoration_amd import ops
    rng = np.random.default_rng(3)
    B, N, C = 1, 65536, 64
    qkv = np.empty((B, N, 3 * C), np.float32)
    qkv[..., :2 * C] = rng.standard_normal((B, N, 2 * C)).astype(np.float32) * 1.5
    vconst = rng.uniform(-2, 2, C).astype(np.float32)
    qkv[..., 2 * C:] = vconst
    qkv[0, 40000, C:C + 16] = qkv[0, 123, :16] * 6        # one dominant key late in the sequence for query 123
    u = vconst.view(np.uint32)
    v_bf16 = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    for bf16 in (True, False):
        out = ops.attention(qkv, 4, bf16=bf16)
        assert np.isfinite(out).all()
        # bf16 path: numerator and denominator use the same rounded p, so the result is the bf16-rounded constant;
        # fp32 path: 65 536 sequential fp32 accumulations per row leave ~2e-4 relative (measured), well inside 1e-3
        want = v_bf16 if bf16 else vconst
        assert (np.abs(out - want) / np.abs(want)).max() < (1e-3 if bf16 else 6e-4)


def test_full_size_sampler_shards_and_codecs(models):
    """Three timesteps of the T=1000 schedule at 256x256: batch shards with the global image offset reproduce the
    unsharded run bit for bit (fp32 mode), and the device JPEG codec equals the host Pillow codec inside the loop."""
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler
    s = DDRMJPEGSampler(models(spec.ARCH_JPEG))
    y = tt(O.compress(synth.synth_images(2, 256, seed=22), 10, "jpeg"))
    kw = dict(steps=1000, start_step=995, num_steps=3, seed=5)
    full = s.sample(y, 10, **kw)
    a = s.sample(y[:1], 10, image_offset=0, **kw)
    b = s.sample(y[1:], 10, image_offset=1, **kw)
    import torch
    assert torch.equal(full, torch.cat([a, b]))
    host = s.sample(y, 10, device_codec=False, **kw)
    assert torch.equal(full, host)
    assert torch.isfinite(full).all() and float(full.abs().max()) < 4.0    # x_t still carries the injected noise here


def test_full_size_forward_matches_oracle_fixture(models):
    """fp32 HIP forward at 256x256 (65 536-token attention at level 0) against the numpy oracle's output for the same
    seeded weights and image (tests/golden/jpeg_fwd256_oracle.npz, made by make_fullsize_oracle.py): north_star's 1e-3."""
    import os
    import torch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_fwd256_oracle.npz"))
    assert int(g["seed_w"]) == 0                       # the `models` fixture loads synth weights with seed 0
    x = tt(synth.synth_images(1, 256, seed=int(g["seed_x"])))
    y = models(spec.ARCH_JPEG)(x, tt(g["t"]), tt(g["cl"])).cpu().numpy()
    err = np.abs(y - g["y"])
    assert err.max() < TOL, (err.max(), err.mean())
    y16 = models(spec.ARCH_JPEG, dtype="bf16")(x, tt(g["t"]), tt(g["cl"])).cpu().numpy()
    e16 = np.abs(y16 - g["y"])
    assert e16.max() < 2e-2 and e16.mean() < 2e-3, (e16.max(), e16.mean())
    print(f"256x256 forward vs oracle: fp32 max {err.max():.2e} mean {err.mean():.2e}; bf16 max {e16.max():.2e} mean {e16.mean():.2e}")


@pytest.mark.parametrize("B", [1, 3, 5])
def test_forward_odd_batch_sizes_bf16(B, models):
    """Odd batches change every GEMM's tiling (split-K decisions, ragged last tiles): each image must still come out as
    it does alone -- bit for bit in fp32 mode, within bf16 re-rounding in bf16 mode."""
    import torch
    x = tt(synth.synth_images(B, 64, seed=40 + B))
    t = torch.linspace(0.1, 0.9, B, device="cuda")
    m32, m16 = models(spec.ARCH_WEBP), models(spec.ARCH_WEBP, dtype="bf16")
    y32, y16 = m32(x, t, t), m16(x, t, t)
    assert torch.isfinite(y16).all()
    e = (y16 - y32).abs()
    assert float(e.max()) < 2e-2 and float(e.mean()) < 2e-3
    for i in range(B):
        assert torch.equal(m32(x[i:i + 1], t[i:i + 1], t[i:i + 1]), y32[i:i + 1])
        d = (m16(x[i:i + 1], t[i:i + 1], t[i:i + 1]) - y16[i:i + 1]).abs()
        assert float(d.max()) < 1e-2 and float(d.mean()) < 1e-3, (i, float(d.max()), float(d.mean()))


def test_full_size_phase_consistency_properties(models):
    """webp_inference.py:531-550 at 256x256 through size-independent properties: with ref = x the corrected image is x
    itself (|X| e^{i arg X} = X); with any ref the spectrum magnitude of the corrected part equals that of x; and the
    result agrees with numpy's FFT (float64) to fp32 DFT accuracy."""
    import torch
    from ddpm_image_restoration_amd.sampler import phase_consistency
    m = models(spec.ARCH_WEBP)
    x = synth.synth_images(2, 256, seed=61)
    ref = synth.synth_images(2, 256, seed=62)
    same = phase_consistency(m, tt(x), tt(x), alpha=0.7).cpu().numpy()
    assert np.abs(same - x).max() < 2e-4
    got = phase_consistency(m, tt(x), tt(ref), alpha=0.7).cpu().numpy()
    X, R = np.fft.fft2(x.astype(np.float64)), np.fft.fft2(ref.astype(np.float64))
    adj = np.real(np.fft.ifft2(np.abs(X) * np.exp(1j * np.angle(R))))
    want = 0.7 * x + (1 - 0.7) * adj
    assert np.abs(got - want).max() < 5e-4
    corrected = (got - 0.7 * x) / 0.3                                   # the ifft2(|X| e^{i arg R}) part alone
    assert np.allclose(np.abs(np.fft.fft2(corrected.astype(np.float64))), np.abs(X), rtol=0, atol=2e-2 * np.abs(X).max() / 100)


@pytest.mark.parametrize("name,arch", [("jpeg", spec.ARCH_JPEG), ("webp", spec.ARCH_WEBP)])
def test_forward_128_against_reference(name, arch, models):
    """The largest size the REFERENCE can run here (16 384-token attention, N x N scores materialised): fp32 HIP forward
    within 1e-3 (measured ~3e-6), bf16 mode within its budget.  Fixture: tests/golden/make_golden_128.py."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{name}_fwd128_ref.npz"))
    x = tt(synth.synth_images(1, 128, seed=int(g["seed_x"])))
    y = models(arch)(x, tt(g["t"]), tt(g["cl"])).cpu().numpy()
    err = np.abs(y - g["y"])
    assert err.max() < TOL, err.max()
    e16 = np.abs(models(arch, dtype="bf16")(x, tt(g["t"]), tt(g["cl"])).cpu().numpy() - g["y"])
    assert e16.max() < 2e-2 and e16.mean() < 2e-3, (e16.max(), e16.mean())
    print(f"{name} 128x128 vs reference: fp32 max {err.max():.2e}; bf16 max {e16.max():.2e} mean {e16.mean():.2e}")


def test_create_destroy_does_not_leak_device_memory(synth_state):
    """Contexts own their weights and workspace: building, running and dropping a model repeatedly leaves the free device
    memory where it was (ddpmir_destroy frees the blob, both plans' arenas, pinned buffers, streams and events)."""
    import gc
    import torch
    from ddpm_image_restoration_amd.model import WebPDiffusionModel
    from ddpm_image_restoration_amd.sampler import DDRMWebPSampler
    sd = synth_state(spec.ARCH_WEBP)
    x = tt(synth.synth_images(2, 64, seed=3))
    t = torch.full((2,), 0.5, device="cuda")

    def cycle():
        m = WebPDiffusionModel(device=torch.device("cuda", 0), dtype="bf16")
        m.load_state_dict(sd)
        m(x, t, t)
        DDRMWebPSampler(m).sample(x, 50, steps=2, seed=1)
        del m
        gc.collect()
        torch.cuda.synchronize()

    cycle()                                             # first cycle: lazy runtime allocations (code objects, pools)
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        cycle()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)     # nothing the size of a weight blob (450 MB+) or an arena is left behind


@pytest.mark.parametrize("H,W", [(160, 160), (32, 224), (96, 32)])
def test_forward_sizes_with_ragged_levels(H, W, models):
    """Sizes whose deep levels are not multiples of the kernel tiles (160 -> 5x5 maps, 25 tokens; 224 x 32 -> 7 x 1): the
    fused epilogues fall back where their alignment rules do not hold; bf16 mode must still agree with fp32 mode, and
    fp32 mode with the oracle."""
    import torch
    x = synth.synth_images(1, max(H, W), seed=70)[:, :, :H, :W].copy()
    t = np.array([0.45], np.float32)
    y32 = models(spec.ARCH_JPEG)(tt(x), tt(t), tt(t))
    y16 = models(spec.ARCH_JPEG, dtype="bf16")(tt(x), tt(t), tt(t))
    e = (y16 - y32).abs()
    assert torch.isfinite(y16).all() and float(e.max()) < 2e-2 and float(e.mean()) < 2e-3, (float(e.max()), float(e.mean()))
    if H * W <= 32 * 224:                                # the oracle takes seconds at these sizes
        from ddpm_image_restoration_amd import synth as S
        ref = O.Model(spec.ARCH_JPEG, S.synth_state_dict(spec.ARCH_JPEG, 0)).forward(x, t, t)
        assert np.abs(y32.cpu().numpy() - ref).max() < TOL


@pytest.mark.parametrize("name,arch", ARCHS)
def test_large_nonsquare_forward_bf16_tracks_fp32(name, arch, models):
    """192 x 320 (61 440 tokens at level 0: not a power of two, 120 query blocks of 512, rows of unequal length at every
    level): the no-running-max attention kernel, the fused qkv packing and the DMA GEMMs against the exact-fp32 path of the
    same library (itself pinned by the oracle at every size the oracle can run)."""
    import torch
    x = synth.synth_images(2, 320, seed=71)[:, :, :192, :].copy()
    t = np.array([0.8, 0.15], np.float32)
    y32 = models(arch)(tt(x), tt(t), tt(t))
    y16 = models(arch, dtype="bf16")(tt(x), tt(t), tt(t))
    e = (y16 - y32).abs()
    print(f"{name} 192x320: bf16 vs fp32 max {float(e.max()):.2e} mean {float(e.mean()):.2e}")
    assert torch.isfinite(y16).all() and float(e.max()) < 2e-2 and float(e.mean()) < 2e-3, (float(e.max()), float(e.mean()))
    # and each image comes out as it does alone (batch 2 -> 1 changes the attention grid and every GEMM's tiling)
    y1 = models(arch)(tt(x[1:]), tt(t[1:]), tt(t[1:]))
    assert torch.equal(y1, y32[1:])


@pytest.mark.parametrize("name,arch", [("webp", spec.ARCH_WEBP), ("jpeg", spec.ARCH_JPEG)])
def test_nonsquare_forward_with_the_bf16_activation_stream(name, arch, models):
    """The bf16 activation stream (fp8 mode's default, bf16 mode's option) on non-square sizes, rows of unequal length at every
    level, both block sizes of the head's mixed transform (WebP 4, JPEG 8): the pool / bilinear / statistics passes and the
    LDS-staged head transform take H and W separately.  Against the exact-fp32 path of the same library, within each mode's
    fence; attention at the bottleneck only (what fp8 mode is specified for, and the level-0 tensors are the large ones)."""
    import torch
    for (H, W) in ((96, 160), (224, 64)):
        x = synth.synth_images(2, max(H, W), seed=73)[:, :, :H, :W].copy()
        t = np.array([0.6, 0.3], np.float32)
        y32 = models(arch, attn_levels="bottleneck")(tt(x), tt(t), tt(t))
        m16 = models(arch, attn_levels="bottleneck", dtype="bf16")
        m16.set_option("bf16_activation_stream", 1)
        e16 = (m16(tt(x), tt(t), tt(t)) - y32).abs()
        m16.set_option("bf16_activation_stream", -1)
        e8 = (models(arch, attn_levels="bottleneck", dtype="fp8")(tt(x), tt(t), tt(t)) - y32).abs()
        print(f"{name} {H}x{W}: bf16 stream max {float(e16.max()):.2e} mean {float(e16.mean()):.2e}; fp8 max {float(e8.max()):.2e} mean {float(e8.mean()):.2e}")
        assert float(e16.max()) < 2.5e-2 and float(e16.mean()) < 2.5e-3, (H, W, float(e16.max()), float(e16.mean()))
        assert float(e8.max()) < 0.17 and float(e8.mean()) < 1.1e-2, (H, W, float(e8.max()), float(e8.mean()))


@pytest.mark.parametrize("name,arch", [("webp", spec.ARCH_WEBP), ("avif", spec.ARCH_AVIF)])
def test_two_lane_loop_shares_the_levels_below_full_resolution(name, arch, models):
    """Host-codec families: the loop runs down1 / up5 per lane and everything below once for both lanes (model.hip fwd_head /
    fwd_mid / fwd_tail).  Images are independent, so a batch of 6 (lanes of 3) must equal its images restored alone or in
    other groupings -- bit for bit in fp32 mode -- including across a phase-fix step and the final step."""
    import torch
    from ddpm_image_restoration_amd import harness
    from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
    x0 = synth.synth_images(6, 64, seed=31)
    y = harness.COMPRESS[arch](torch.from_numpy(x0), 10).cuda()
    s32 = SAMPLER_BY_ARCH[arch](models(arch))
    kw = dict(steps=7, seed=9)                          # i = 6 .. 0: phase fixes at i = 6, 3 (AVIF) / 5 (WebP), last step i = 0
    full = s32.sample(y, 10, **kw)
    parts = [s32.sample(y[a:b], 10, image_offset=a, **kw) for a, b in ((0, 2), (2, 3), (3, 6))]   # lanes of 1, no lanes, odd split
    assert torch.equal(full, torch.cat(parts))
    s16 = SAMPLER_BY_ARCH[arch](models(arch, dtype="bf16"))
    f16 = s16.sample(y, 10, **kw)
    assert torch.isfinite(f16).all()
    # bf16 mode: same images through another grouping stay within the loop's bf16 scatter (GEMM tilings differ with the batch)
    g16 = torch.cat([s16.sample(y[a:b], 10, image_offset=a, **kw) for a, b in ((0, 4), (4, 6))])
    assert float((f16 - g16).abs().mean()) < 5e-2


@pytest.mark.parametrize("dtype,size,batch,steps,q,mode", [("bf16", 64, 6, 7, 10, 1), ("fp8", 64, 4, 6, 30, 2), ("bf16", 128, 4, 6, 10, 2)])
def test_two_concurrent_lanes_equal_the_lanes_run_one_after_the_other(dtype, size, batch, steps, q, mode, synth_state):
    """Option concurrent_lanes (device-codec JPEG loop, bf16 / fp8 mode): the batch is split into two lanes that run concurrently
    on two streams -- this context and a second one on the same weight blob -- with their full-resolution attention launches
    taking turns.  Each lane must produce EXACTLY what the single-lane loop produces for that half of the batch on its own
    (same launch batch -> same kernel selection; the noise is keyed by the global image index), including the FFT phase fix
    (q 10, every 5th index), slices of the schedule (start_step / num_steps) and repeated calls; profiling counters add up over
    both lanes."""
    import torch
    from ddpm_image_restoration_amd import harness
    from ddpm_image_restoration_amd.model import JPEGDiffusionModel
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler
    dev = torch.device("cuda", 0)
    sd = synth_state(spec.ARCH_JPEG)
    m = JPEGDiffusionModel(device=dev, dtype=dtype)
    m.load_state_dict(sd)
    s = DDRMJPEGSampler(m)
    x0 = synth.synth_images(batch, size, seed=91)
    y = harness.COMPRESS[spec.ARCH_JPEG](torch.from_numpy(x0), q).cuda()
    h = batch // 2
    m.set_option("concurrent_lanes", 0)
    halves = torch.cat([s.sample(y[:h], q, steps=steps, seed=5, image_offset=40),
                        s.sample(y[h:], q, steps=steps, seed=5, image_offset=40 + h)])
    whole = s.sample(y, q, steps=steps, seed=5, image_offset=40)
    m.set_option("concurrent_lanes", mode)
    m.profile(True)
    lanes = s.sample(y, q, steps=steps, seed=5, image_offset=40)
    m.profile(False)
    assert torch.isfinite(lanes).all()
    assert torch.equal(lanes, halves), float((lanes - halves).abs().max())
    a_ms, a_n, a_fl = m.profile_read("attention_l0")
    f_ms, f_n, _ = m.profile_read("forward")
    assert a_n == 4 * steps and f_n == 2 * steps and a_ms > 0                       # two launches per lane and step; one forward per lane and step
    assert abs(a_fl - 4 * steps * 4.0 * (size * size) ** 2 * 64 * h) < 1e-6 * a_fl
    assert torch.equal(s.sample(y, q, steps=steps, seed=5, image_offset=40), lanes)  # again: same bits
    # (the unsplit batch is a different launch batch: other kernel choices, and the codec's quantisation amplifies the mode's noise floor
    #  from step to step -- the same restoration, not the same bits)
    assert float((lanes - whole).abs().mean()) < 0.2
    # a slice of the schedule, then the rest: the lanes continue from x_t like the single-lane loop
    a = s.sample(y, q, steps=steps, seed=5, start_step=steps - 1, num_steps=3)
    m.set_option("concurrent_lanes", 0)
    b = torch.cat([s.sample(y[:h], q, steps=steps, seed=5, start_step=steps - 1, num_steps=3),
                   s.sample(y[h:], q, steps=steps, seed=5, start_step=steps - 1, num_steps=3, image_offset=h)])
    assert torch.equal(a, b)
    # odd batches and injected noise stay on the single-lane loop (and still work with the option on)
    m.set_option("concurrent_lanes", 1)
    odd = s.sample(y[:3], q, steps=3, seed=1)
    m.set_option("concurrent_lanes", 0)
    assert torch.equal(odd, s.sample(y[:3], q, steps=3, seed=1))


@pytest.mark.parametrize("arch,dtype,stream16", [(spec.ARCH_JPEG, "bf16", -1), (spec.ARCH_JPEG, "bf16", 1), (spec.ARCH_JPEG, "fp8", -1), (spec.ARCH_WEBP, "bf16", -1),
                                                 (spec.ARCH_AVIF, "bf16", -1)])
def test_output_head_in_one_pass_equals_the_three_launches(arch, dtype, stream16, synth_state):
    """head_fused.hip: GroupNorm-apply + SiLU + conv 64 -> 3 + tanh + NCHW as ONE kernel (halo tile through registers into LDS, one
    MFMA chain per tile row) against the generic path it replaces (gn_apply16 pass, strip conv with the 3 outputs padded to 32
    columns, layout pass): the same bf16 operand and the same products, fp32 accumulation in another order -> agreement to a few
    ulps of the pre-tanh sum; image borders (zero padding of the ACTIVATED tensor), fp32 and bf16 input (the bf16 activation
    stream), several tiles per image and a non-square image."""
    import torch
    from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
    m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), dtype=dtype, attn_levels="bottleneck")
    if stream16 >= 0:
        m.set_option("bf16_activation_stream", stream16)
    m.load_state_dict(synth_state(arch))
    for (B, H, W) in ((2, 32, 32), (3, 64, 96), (1, 128, 64)):
        x = torch.from_numpy(synth.synth_images(B, max(H, W), seed=17)[:, :, :H, :W].copy()).cuda()
        t = torch.linspace(0.1, 0.9, B, device="cuda")
        m.set_option("head_fused", 1)
        a = m(x, t, t)
        assert torch.equal(m(x, t, t), a)
        m.set_option("head_fused", 0)
        b = m(x, t, t)
        d = float((a - b).abs().max())
        assert torch.isfinite(a).all() and d < 2e-5, (arch, dtype, (B, H, W), d)
        assert float(a.abs().max()) > 1e-3


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_groupnorm_apply_fused_into_the_full_resolution_conv_in_the_model(dtype, synth_state):
    """option conv_tile = 2 routes norm2 + GELU + conv2 of the 64-channel blocks through conv_tile.hip (persistent,
    weights-stationary, loader waves + MFMA waves) instead of the operand pass + strip kernel.  Same operand values and products,
    another accumulation order: the forward moves by bf16 mode's noise floor at most (a last-bit difference of conv2's fp32 sums
    re-draws some bf16 roundings downstream), and is deterministic."""
    import torch
    from ddpm_image_restoration_amd.model import JPEGDiffusionModel
    m = JPEGDiffusionModel(device=torch.device("cuda", 0), dtype=dtype, attn_levels="bottleneck")
    m.load_state_dict(synth_state(spec.ARCH_JPEG))
    for (B, R) in ((2, 64), (1, 128)):
        x = torch.from_numpy(synth.synth_images(B, R, seed=23)).cuda()
        t = torch.full((B,), 0.35, device="cuda")
        m.set_option("conv_tile", 0)
        a = m(x, t, t)
        m.set_option("conv_tile", 2)
        b = m(x, t, t)
        assert torch.equal(m(x, t, t), b)
        d = (a - b).abs()
        # (fp8 mode: e4m3 roundings downstream are re-drawn too -- its own budget against the oracle is max 0.17, mean 1.1e-2)
        fmean, fmax = (1.5e-3, 2e-2) if dtype == "bf16" else (1.1e-2, 0.17)
        assert torch.isfinite(b).all() and float(d.mean()) < fmean and float(d.max()) < fmax, (dtype, R, float(d.mean()), float(d.max()))
        assert not torch.equal(a, b) or R < 64
    m.set_option("conv_tile", 1)

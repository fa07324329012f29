#!/usr/bin/env python3
"""Headline benchmark: restored images/sec of the DDRM reverse-diffusion sampler on MI355X.

Default workload (BASELINE.json configs[1]): JPEG restoration U-Net (reference architecture: self-attention over all
H*W tokens in every one of the 13 blocks), 256x256, batch 16 per GPU, T=1000, JPEG q10, synthetic JPEG-degraded inputs
and synthetic weights (the reference ships no checkpoint).  `--arch webp` (T=1000) and `--arch avif --T 250` are the
per-GPU shapes of configs[2] and configs[3]; they run the host codec of their family inside the loop.

A "step" is ONE reverse-diffusion timestep over the batch: U-Net forward + uint8 quantise + codec round trip + DDRM
update (+ the FFT phase fix on every 5th / 3rd index at q10) -- webp_inference.py:566-600.  All T timesteps do identical
work, so  images/sec = N_gpus * B / (T * seconds_per_step); the K timed steps are consecutive indices of the real T-step
schedule run through the product C ABI (ddpmir_ddrm_sample_loop with start_step/num_steps).  `--full` runs the whole
T-step restoration instead (minutes).

The LAST stdout line is the one JSON object of the contract and stays below LINE_LIMIT characters (the driver keeps an
8 000-character tail of stdout + stderr; round 3's 24 KB line was cut in half there).  The headline keys are those of
configs[1].  At N=1 with the default workload, AFTER the headline has been measured, a fresh child process runs short
slices of the other BASELINE configurations (configs[2] and [3] per-GPU shapes, configs[4] shape in bf16 and fp8 mode,
configs[0] in full, the three shapes the reference's notebooks left a tqdm rate for, and the headline workload with the
no-max attention kernel disabled = its worst case): a device fault in a slice cannot take the headline with it.  Their
full records (each with its own roofline object) go to gpurun_out/bench_workloads.json; the line carries the compact
`workloads_summary` (id, ms_per_step, images_per_sec, roofline_frac).  `--no-workloads` skips them.

  python bench.py                       # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (fp8 mode: the attention kernels stay bf16; the e4m3 GEMM launches
# are priced against the block-scaled fp8 peak)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp8": 2500.0}
PEAK_FP8_TFLOPS = 5000.0
NOISE_COEFF = {"jpeg": 0.2, "webp": 0.2, "avif": 0.15}
PHASE = {"jpeg": (20, 5, 0.7), "webp": (15, 5, 0.7), "avif": (30, 3, 0.8)}     # quality threshold, period, alpha
# algorithmic GFLOP per image and forward (attention in all 13 blocks / at the bottleneck only)
STEP_GFLOP = {("all", 32): 2.62, ("all", 64): 17.61, ("all", 256): 2564.0, ("all", 512): 39470.0,
              ("bottleneck", 256): 116.2, ("bottleneck", 512): 465.4}
STEP_GFLOP_AVIF = {("all", 32): 3.25, ("all", 64): 20.14, ("all", 256): 2604.5, ("all", 512): 39632.0,
                   ("bottleneck", 256): 156.7, ("bottleneck", 512): 627.2}


def cpu_baseline(arch_name: str, size: int, quality: int, T: int, n_steps: int = 1, batch: int = 1):
    """numpy oracle ("port") timed on this host's cores on a bounded sample of the same workload:
    `batch` images (default 1), `n_steps` timesteps of the T-step schedule at the full image size."""
    import numpy as np
    from ddpm_image_restoration_amd import spec, synth
    from oracle import ddpm_oracle as O
    arch = spec.ARCH_NAMES[arch_name]
    sd = synth.synth_state_dict(arch)
    m = O.Model(arch, sd)
    x0 = synth.synth_images(batch, size, seed=123)
    y = O.compress(x0, quality, arch_name)
    x_t = y.copy()
    pq, pevery, palpha = PHASE[arch_name]
    t0 = time.perf_counter()
    for i in range(T - 1, T - 1 - n_steps, -1):
        t = (np.full((batch,), i, dtype=np.float32) / np.float32(T)).astype(np.float32)
        th = m(x_t, t, t)
        dec = O.compress(th, quality, arch_name)
        xp = (th - dec) + y
        eps = np.random.default_rng(i).standard_normal(x_t.shape).astype(np.float32)
        x_t = (xp + np.float32(0.85) * (eps * (t * np.float32(NOISE_COEFF[arch_name])).reshape(-1, 1, 1, 1))).astype(np.float32)
        if quality < pq and i % pevery == 0:
            x_t = O.phase_consistency(x_t, y, palpha)
    dt = (time.perf_counter() - t0) / n_steps
    threads = O.oracle_threads()    # the oracle's attention fans out over this many threads (BLAS single-threaded inside)
    return {"value": batch / (dt * T), "unit": "images/sec", "cores": int(threads), "kind": "port",
            "sample": f"numpy oracle, {arch_name} U-Net, B={batch} {size}x{size}, {n_steps} of {T} timesteps, {dt:.2f} s/step"}


def committed_traffic(key: str, lib_src_hash: str | None = None):
    """HBM-side bytes per launch of the roofline kernel from the newest committed PMC summary (profiles/*traffic*.json,
    written by tools/pmc_summary.py from two `rocprofv3 --pmc` passes over this very command: FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE).  rocprofv3 counters cannot be read from inside the run, so the value carries its source --
    and `stale`: whether the kernels were rebuilt from other sources since (the library's source hash, ddpmir_version())."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        e = d.get("workloads", {}).get(key)
        if e:
            best = (e, os.path.relpath(f, ROOT), d.get("commit"))
    if not best:
        return None, None
    e, src, commit = best
    measured_hash = e.get("src_hash")
    stale = None if (measured_hash is None or lib_src_hash is None) else (measured_hash != lib_src_hash)
    return float(e["roofline_kernel_bytes_per_launch"]), {"file": src, "commit": commit, "kernel": e.get("roofline_kernel"),
                                                           "command": e.get("command"), "src_hash": measured_hash,
                                                           "stale": True if measured_hash is None else stale}


def lib_src_hash(version: str):
    for tok in version.split():
        if tok.startswith("src="):
            return tok[4:]
    return None


def run_workload(w: dict, env, dev, K: int, W: int, full: bool = False, graph: int = 0, host_codec: str = "native",
                 codec: str | None = None, detail_pass: bool = True):
    """One workload: build the model of its family / mode, load synthetic weights (rank 0 packs, RCCL broadcasts), degrade a
    synthetic batch with the family's codec, run W untimed + K timed steps of its T-step schedule through the product's
    sampling loop, and read the HIP-event profile classes.  Returns (summary dict, timing/profile details)."""
    import torch
    from ddpm_image_restoration_amd import dist as D
    from ddpm_image_restoration_amd import _lib, harness, spec, synth
    from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
    from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH

    arch_name, R, B, T, q, attn, dtype = w["arch"], w["size"], w["batch"], w["T"], w["quality"], w["attn"], w["dtype"]
    arch = spec.ARCH_NAMES[arch_name]
    model = MODEL_BY_ARCH[arch](device=dev, dtype=dtype, attn_levels=attn, use_graph=bool(graph), fp8_min_channels=w.get("fp8_min_channels"))
    if w.get("nm_bound") is not None:
        model.set_option("attention_no_max_bound", w["nm_bound"])
    if w.get("stream16") is not None:
        model.set_option("bf16_activation_stream", w["stream16"])
    t_b0 = time.perf_counter()
    D.load_weights_broadcast(model, lambda: synth.synth_state_dict(arch), env)
    torch.cuda.synchronize(dev)
    load_s = time.perf_counter() - t_b0
    sampler = SAMPLER_BY_ARCH[arch](model)

    # synthetic degraded batch of this rank's shard (global image index = rank*B + j): y = codec(x0, q), the product's own codec
    n_total = B * env.world
    lo, hi = D.shard_range(n_total, env.rank, env.world)
    x0 = synth.synth_images(n_total, R, seed=123)[lo:hi]
    y = harness.COMPRESS[arch](torch.from_numpy(x0), q).to(dev)
    dev_codec = arch_name == "jpeg" and (codec or "device") == "device"

    first = T - 1
    common = dict(steps=T, seed=7, image_offset=lo, device_codec=dev_codec, host_codec=host_codec)
    if full:
        K, W = 1, 0
    K = min(K, T)
    W = min(W, max(0, T - K))
    if W > 0:
        sampler.sample(y, q, start_step=first, num_steps=W, **common)
    torch.cuda.synchronize(dev)
    # HIP-event spans force eager launches (events cannot bracket nodes of a replayed graph): with --graph 1 the timed region
    # replays the captured timesteps and the line carries no roofline object
    model.profile(not graph)
    D.barrier(env)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if full:
        sampler.sample(y, q, **common)
    else:
        sampler.sample(y, q, start_step=first - W, num_steps=K, **common)
    torch.cuda.synchronize(dev)
    D.barrier(env)
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, env, device=dev)
    model.profile(False)

    steps_timed = T if full else K
    ms_per_step = dt * 1e3 / steps_timed
    a_ms, a_n, a_fl = model.profile_read("attention_l0")
    f_ms, f_n, _ = model.profile_read("forward")
    c_ms, c_n, _ = model.profile_read("codec_host")
    _, g_taken, g_total = model.profile_read("attention_general")
    det = {"ms_per_step": ms_per_step, "steps_timed": steps_timed, "warmup": W, "n_total": n_total, "load_s": load_s,
           "dev_codec": dev_codec, "version": model.version(), "weight_blob_bytes": model.weight_blob_bytes(),
           "codec_threads": (None if dev_codec else int(_lib.load().ddpmir_host_codec_threads()))}
    heads = 8 if arch_name == "avif" else 4
    peak = PEAK_TFLOPS[dtype]
    roof = None
    if a_n > 0 and a_ms > 0 and attn == "all":
        achieved = a_fl / (a_ms * 1e-3) / 1e12
        traffic, tsrc = committed_traffic(f"{arch_name}_{dtype}_{R}_B{B}_{attn}", lib_src_hash(det["version"]))
        roof = {"bound": "mfma", "kernel": "flash attention of the full-resolution blocks (down1/up5): N=%d tokens, %d heads, head_dim %d"
                                           % (R * R, heads, 64 // heads),
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": traffic, "traffic_source": tsrc,
                "binding_resource": ("VALU issue (one v_exp_f32 per score, %d FLOP per exp); priced against the bf16 MFMA peak -- DESIGN.md 4.1" % (4 * 64 // heads)
                                     if dtype != "f32" else "exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 tiles)"),
                "avg_launch_ms": a_ms / a_n, "launches": a_n, "forward_ms": (f_ms / f_n) if f_n else None}
    codec_obj = None
    if not dev_codec and c_n > 0:
        # host time inside the codec (summed over the lanes of a step) next to the GPU time of the step's forwards: what
        # the two-lane pipeline cannot hide is ms_per_step - forwards - the step's small device kernels.  Max over ranks:
        # on a multi-GPU host the ranks' codec pools share the cores, so the slowest rank is the one that binds.
        fwd_per_step = f_ms / steps_timed if f_n else None
        not_hidden = max(0.0, ms_per_step - fwd_per_step) if fwd_per_step else 0.0
        codec_obj = {"host_ms_per_step": D.max_over_ranks(c_ms / steps_timed, env, device=dev), "calls_per_step": c_n / steps_timed,
                     "forward_ms_per_step": fwd_per_step,
                     "not_hidden_ms_per_step": D.max_over_ranks(not_hidden, env, device=dev), "over": "max over ranks"}
    gemm = None
    if detail_pass and dtype != "f32" and not graph:
        # GEMM-class roofline from a separate, untimed pass (about 220 event records per forward would disturb the timed one):
        # every launch of the DMA-streamed GEMM kernels (3x3 / 1x1 convs, projections) bracketed with HIP events on the library's stream
        model.profile(True, detail=3)
        n_d = min(2, max(1, T - W - steps_timed))
        sampler.sample(y, q, start_step=max(n_d - 1, first - W - steps_timed), num_steps=n_d, **common)
        torch.cuda.synchronize(dev)
        model.profile(False)
        gm, gn, gf = model.profile_read("gemm")
        g8m, g8n, g8f = model.profile_read("gemm_fp8")
        fd_ms, fd_n, _ = model.profile_read("forward")
        if gn > 0 and gm > 0:
            gemm = {"launches_per_forward": (gn + g8n) / max(1, fd_n), "ms_per_forward": (gm + g8m) / max(1, fd_n),
                    "bf16": {"achieved": gf / (gm * 1e-3) / 1e12, "peak": PEAK_TFLOPS["bf16"], "frac": gf / (gm * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
                             "launches": gn, "ms": gm, "gflop": gf / 1e9}}
            if g8n > 0 and g8m > 0:
                gemm["fp8"] = {"achieved": g8f / (g8m * 1e-3) / 1e12, "peak": PEAK_FP8_TFLOPS, "frac": g8f / (g8m * 1e-3) / 1e12 / PEAK_FP8_TFLOPS,
                               "launches": g8n, "ms": g8m, "gflop": g8f / 1e9}
            tot_fl, tot_ms = gf + g8f, gm + g8m
            gemm["all"] = {"achieved": tot_fl / (tot_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                           "frac_of_bf16_peak": tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"]}
    if roof is None and gemm is not None:
        # attention at the bottleneck only: the GEMM launches are the dominant kernel class of the timestep
        fmc, s16 = w.get("fp8_min_channels"), w.get("stream16")
        variant = dtype + ("@%d" % fmc if fmc not in (None, 256) else "") + ("+stream16" if s16 == 1 else "")
        traffic, tsrc = committed_traffic(f"{arch_name}_{variant}_{R}_B{B}_{attn}", lib_src_hash(det["version"]))
        roof = {"bound": "mfma", "kernel": "GEMM class: all %d launches per forward of the DMA-streamed implicit-GEMM kernels (3x3 / 1x1 convolutions, "
                                           "projections; gemm16_kernel / gemm256_kernel / conv3x3_strip_kernel / conv_tile_kernel / head_fused_k / splitk_reduce_k)" % round(gemm["launches_per_forward"]),
                "achieved": gemm["all"]["achieved"], "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "frac": gemm["all"]["frac_of_bf16_peak"],
                "traffic": traffic, "traffic_source": tsrc, "per_launch": "one forward's GEMM launches (sum of 2*M*N*K / sum of event times)",
                "gemm_ms_per_forward": gemm["ms_per_forward"], "bf16_launches": gemm["bf16"],
                "fp8_launches": gemm.get("fp8"), "measured_in": "separate untimed pass with per-launch HIP events"}
    table = STEP_GFLOP_AVIF if arch_name == "avif" else STEP_GFLOP
    gfl = table.get((attn, R))
    summary = {"name": w.get("name"), "arch": arch_name, "image": R, "batch_per_gpu": B, "T": T, "quality": q, "attn_levels": attn,
               "dtype": dtype, "steps": steps_timed, "warmup": W, "ms_per_step": ms_per_step,
               "steps_per_sec": 1e3 / ms_per_step, "images_per_sec": n_total / (T * ms_per_step / 1e3),
               "whole_step_tflops": (gfl * 1e9 * n_total / (ms_per_step * 1e-3) / 1e12) if gfl else None,
               "roofline": roof, "gemm_class": gemm if roof is None or roof.get("bf16_launches") is None else None,
               "attention_routing": ({"general_kernel_workgroups_with_work": g_taken, "general_kernel_workgroups_launched": int(g_total),
                                      "share_general": (g_taken / g_total) if g_total else None,
                                      "meaning": "query blocks whose score bound exceeds 80 (log2) take the running-max kernel"}
                                     if g_total else None),
               "codec": codec_obj}
    if w.get("reference_it_per_s"):
        summary["reference_it_per_s"] = w["reference_it_per_s"]
        summary["reference_source"] = w.get("reference_source")
    if w.get("reference_cpu_images_per_sec"):
        summary["reference_cpu_images_per_sec"] = w["reference_cpu_images_per_sec"]
        summary["reference_source"] = w.get("reference_source")
    if w.get("cpu_sample_steps") and env.world == 1:   # the oracle on this host's cores, on a bounded sample of THIS workload
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-steps", "--size", str(R), "--quality", str(q),
                                "--arch", arch_name, "--T", str(T), "--batch", str(B), "--steps", str(w["cpu_sample_steps"])],
                               capture_output=True, text=True, timeout=300)
            out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            summary["cpu_baseline"] = json.loads(out[-1]) if (r.returncode == 0 and out) else {"value": None, "sample": "failed: " + r.stderr.strip()[-200:]}
        except Exception as e:
            summary["cpu_baseline"] = {"value": None, "sample": "not measured: %r" % (e,)}
    if w.get("note"):
        summary["note"] = w["note"]
    det.update(roof=roof, codec=codec_obj, gemm=gemm, routing=summary["attention_routing"])
    del sampler, model
    torch.cuda.empty_cache()
    return summary, det


EXTRA_WORKLOADS = [
    dict(id="c2_webp", name="BASELINE configs[2] per-GPU shape: WebP 256x256, B=16, T=1000, q10 (host libwebp pool, two lanes)",
         arch="webp", size=256, batch=16, T=1000, quality=10, attn="all", dtype="bf16", steps=5, warmup=2),
    dict(id="c3_avif", name="BASELINE configs[3] per-GPU shape: AVIF 256x256, B=16, T=250, q10 (host libavif pool, two lanes)",
         arch="avif", size=256, batch=16, T=250, quality=10, attn="all", dtype="bf16", steps=4, warmup=2),
    dict(id="c4_bf16", name="BASELINE configs[4] shape in bf16 mode: JPEG 512x512, B=8, T=1000, q20, attention at the bottleneck only",
         arch="jpeg", size=512, batch=8, T=1000, quality=20, attn="bottleneck", dtype="bf16", steps=10, warmup=3),
    dict(id="c4_bf16_s16", name="BASELINE configs[4] shape in bf16 mode with the bf16 activation stream (option bf16_activation_stream = 1: tensors between "
                                "residual blocks stored as bf16, fp8 mode's default; opt-in in bf16 mode: forward error mean 2.0e-3 vs 1.7e-3, restored PSNR unchanged)",
         arch="jpeg", size=512, batch=8, T=1000, quality=20, attn="bottleneck", dtype="bf16", steps=10, warmup=3, stream16=1),
    dict(id="c4_fp8", name="BASELINE configs[4] shape, fp8 mode (e4m3 operands in the 3x3 convs with >= 256 channels)",
         arch="jpeg", size=512, batch=8, T=1000, quality=20, attn="bottleneck", dtype="fp8", steps=10, warmup=3),
    dict(id="c4_fp8_64", name="BASELINE configs[4] shape, fp8 mode with e4m3 operands in every 3x3 conv with >= 64 channels (fp8_min_channels = 64; "
                              "parity fences: tests/test_gpu_fullsize.py::test_fp8_min_channels_model_level)",
         arch="jpeg", size=512, batch=8, T=1000, quality=20, attn="bottleneck", dtype="fp8", steps=10, warmup=3, fp8_min_channels=64),
    dict(id="c0_f32", name="BASELINE configs[0] on the GPU: JPEG 32x32, B=4, T=50, q10, complete restoration (fp32 parity mode)",
         arch="jpeg", size=32, batch=4, T=50, quality=10, attn="all", dtype="f32", steps=45, warmup=5, cpu_sample_steps=5,
         reference_cpu_images_per_sec=0.364,
         reference_source="the reference's own PyTorch CPU path on this configuration: 10.98 s for the batch, 8 threads, survey container (BASELINE.md section 2)"),
    dict(id="c0_bf16", name="BASELINE configs[0] on the GPU, bf16 mode",
         arch="jpeg", size=32, batch=4, T=50, quality=10, attn="all", dtype="bf16", steps=45, warmup=5),
    dict(id="nb_jpeg64", name="reference notebook shape: JPEG 64x64, B=18, 80 steps (svd imagenet.ipynb c0 output: 18.2 it/s on the author's CUDA GPU)",
         arch="jpeg", size=64, batch=18, T=80, quality=20, attn="all", dtype="bf16", steps=60, warmup=10,
         reference_it_per_s=18.2, reference_source="svd imagenet.ipynb c0 output (BASELINE.md section 1); unnamed CUDA GPU"),
    dict(id="nb_webp64", name="reference notebook shape: WebP 64x64, B=18, 80 steps (webp.ipynb c0 output: 16.2 it/s)",
         arch="webp", size=64, batch=18, T=80, quality=20, attn="all", dtype="bf16", steps=60, warmup=10,
         reference_it_per_s=16.2, reference_source="webp.ipynb c0 output (BASELINE.md section 1); unnamed CUDA GPU"),
    dict(id="nb_jpeg32", name="reference notebook shape: JPEG 32x32, B=128, 80 steps (svd.ipynb c0 output: 13.8 it/s)",
         arch="jpeg", size=32, batch=128, T=80, quality=20, attn="all", dtype="bf16", steps=60, warmup=10,
         reference_it_per_s=13.8, reference_source="svd.ipynb c0 output (BASELINE.md section 1); unnamed CUDA GPU"),
    dict(id="worst_attn", name="headline workload, worst case of the attention routing: no-max kernel disabled (every query block takes the "
                               "general running-max kernel, as with unbounded scores everywhere)",
         arch="jpeg", size=256, batch=16, T=1000, quality=10, attn="all", dtype="bf16", steps=4, warmup=1, nm_bound=-1.0,
         note="general_kernel_only"),
]

LINE_LIMIT = 6000          # characters of the final stdout line (the driver's tail of stdout + stderr is 8 000)
WORKLOADS_FILE = os.path.join(ROOT, "gpurun_out", "bench_workloads.json")
# keys the line may lose, in this order, if it ever grows past LINE_LIMIT (the contract's keys are never dropped)
OPTIONAL_KEYS = ["workloads_file", "codec", "attention_routing", "general_kernel_only", "workloads_summary", "distributed", "library"]


def summarise_workloads(extras: list) -> list:
    """Compact form of the extra slices for the headline line: short id, ms per step, images/s, roofline fraction."""
    out = []
    for e in extras:
        if "error" in e or "ms_per_step" not in e:
            out.append({"id": e.get("id"), "error": str(e.get("error", "no result"))[:80]})
            continue
        roof = e.get("roofline") or {}
        s = {"id": e.get("id"), "ms_per_step": round(e["ms_per_step"], 3), "images_per_sec": float("%.4g" % e["images_per_sec"]),
             "roofline_frac": round(roof["frac"], 4) if roof.get("frac") is not None else None}
        if e.get("reference_it_per_s"):
            s["steps_per_sec"] = round(e["steps_per_sec"], 1)
            s["reference_it_per_s"] = e["reference_it_per_s"]
        out.append(s)
    return out


def fit_line(line: dict, limit: int = LINE_LIMIT) -> str:
    """json.dumps(line), dropping optional keys (never the contract's) until it fits the driver's tail."""
    line = dict(line)
    txt = json.dumps(line)
    for k in OPTIONAL_KEYS:
        if len(txt) <= limit:
            break
        if line.pop(k, None) is not None:
            line["dropped_for_length"] = line.get("dropped_for_length", []) + [k]
            txt = json.dumps(line)
    if len(txt) > limit:        # last resort: shorten the free-text fields
        for obj in (line.get("config", {}), line.get("roofline") or {}, line.get("cpu_baseline") or {}):
            for k, v in list(obj.items()):
                if isinstance(v, str) and len(v) > 120:
                    obj[k] = v[:117] + "..."
        txt = json.dumps(line)
    return txt


def headline_line(args, env_world: int, backend: str | None, det: dict, local_world: int, codec_threads: int | None) -> dict:
    """The contract's JSON object from the measured details of the headline workload (pure: no GPU, used by a CPU test too)."""
    T, R, B, q = args.T, args.size, args.batch, args.quality
    ms_per_step, steps_timed, n_total, dev_codec = det["ms_per_step"], det["steps_timed"], det["n_total"], det["dev_codec"]
    cfg_no = {("jpeg", 1000): 1, ("webp", 1000): 2, ("avif", 250): 3}.get((args.arch, T))
    is_named = cfg_no is not None and (R, B, q, args.attn, args.dtype) == (256, 16, 10, "all", "bf16")
    codec_txt = ("device JPEG round trip (bit-exact to libjpeg-turbo)" if dev_codec else
                 "host Pillow/libjpeg-turbo callback" if args.arch == "jpeg" else
                 ("lib%s C-API thread pool (bit-identical to Pillow), two pipelined micro-batch lanes" % args.arch
                  if args.host_codec == "native" else "callback into Pillow, two pipelined micro-batch lanes"))
    line = {
        "metric": "restored images/sec (%dx%d, T=%d)" % (R, R, T), "value": n_total / (T * ms_per_step / 1e3), "unit": "images/sec",
        "n_gpus": env_world, "steps": steps_timed, "warmup": det["warmup"], "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": "%s: %s restoration U-Net, %dx%d, batch %d/GPU, T=%d, %s q%d; "
                               "step = one reverse-diffusion timestep (U-Net forward + codec round trip + DDRM update)"
                               % (("BASELINE configs[%d]%s" % (cfg_no, "" if cfg_no == 1 else " (per-GPU shape)")) if is_named
                                  else "variant of BASELINE configs", args.arch.upper(), R, R, B, T, args.arch.upper(), q),
                   "global_batch": n_total, "image": R, "T": T, "quality": q, "attn_levels": args.attn,
                   "codec": codec_txt, "timestep_launch": "captured hipGraphs" if args.graph else "eager",
                   "arithmetic": {"bf16": "bf16 MFMA operands, fp32 accumulation / softmax / GroupNorm statistics / residual stream; restored PSNR equals "
                                          "fp32 mode's as a mean over an image set (tests/test_gpu_psnr256.py); 1e-3 pixel parity is fp32 mode's (--dtype f32)",
                                  "fp8": "as bf16, with OCP e4m3 operands in the 3x3 convs with >= %d channels" % (args.fp8_min_channels or 256),
                                  "f32": "fp32 matrix cores"}[args.dtype],
                   "parallelism": f"dp{env_world} (independent image shards, no collective in the loop)"},
        "library": det["version"],
        "distributed": {"backend": backend if env_world > 1 else None, "world_size": env_world, "local_world_size": local_world,
                        "codec_threads": codec_threads, "weight_blob_bytes": det["weight_blob_bytes"],
                        "load_pack_broadcast_s": round(det["load_s"], 3)},
    }
    if det.get("codec"):
        line["codec"] = det["codec"]
    if det.get("roof"):
        line["roofline"] = det["roof"]
    if det.get("routing"):
        line["attention_routing"] = det["routing"]
    return line


def workloads_child(args):
    """Child process of the default N=1 run: the extra workload slices, one after the other, the full records rewritten to
    `args.workloads_child` after every slice (so whatever was measured survives a fault in a later one)."""
    import torch
    from ddpm_image_restoration_amd import dist as D
    env = D.DistEnv(0, 1, 0, "none")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sel = [int(i) for i in args.workloads.split(",")] if args.workloads else range(len(EXTRA_WORKLOADS))
    extras = []
    for i in sel:
        w = EXTRA_WORKLOADS[i]
        t0 = time.perf_counter()
        try:
            s, _ = run_workload(w, env, dev, w["steps"], w["warmup"])
            s["id"] = w["id"]
        except Exception as e:
            s = {"id": w["id"], "name": w["name"], "error": repr(e)[:300]}
        s["wall_s"] = round(time.perf_counter() - t0, 2)
        extras.append(s)
        tmp = args.workloads_child + ".tmp"
        with open(tmp, "w") as f:
            json.dump({"workloads": extras}, f, indent=1)
        os.replace(tmp, args.workloads_child)


def run_children(args, line: dict, extra_workloads: bool, cpu: bool):
    """The two child processes of the N=1 run, started only after the headline has been measured and the parent's model freed:
    (1) the extra workload slices on the GPU, (2) the numpy oracle on the host cores.  Their output never reaches this process's
    stdout / stderr (captured into gpurun_out/), so nothing but the final line is printed."""
    import subprocess
    me = os.path.abspath(__file__)
    wfile = WORKLOADS_FILE
    try:
        os.makedirs(os.path.dirname(wfile), exist_ok=True)
        with open(wfile + ".probe", "w"):
            pass
        os.remove(wfile + ".probe")
    except OSError:                     # read-only checkout: the side file goes to the temp directory instead
        import tempfile
        wfile = os.path.join(tempfile.gettempdir(), "bench_workloads.json")
    if extra_workloads:
        try:
            os.remove(wfile)
        except OSError:
            pass
        cmd = [sys.executable, me, "--workloads-child", wfile] + (["--workloads", args.workloads] if args.workloads else [])
        rc, note = None, None
        try:
            with open(wfile + ".log", "w") as log:
                rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, timeout=900).returncode
        except Exception as e:
            note = "%r" % (e,)
        extras = []
        try:
            extras = json.load(open(wfile))["workloads"]
        except Exception:
            pass
        line["workloads_summary"] = summarise_workloads(extras)
        line["workloads_file"] = {"path": os.path.relpath(wfile, ROOT) if wfile.startswith(ROOT) else wfile, "child_rc": rc, "slices": len(extras)}
        if note:
            line["workloads_file"]["note"] = note[:120]
        worst = next((e for e in extras if e.get("note") == "general_kernel_only" and "ms_per_step" in e), None)
        if worst:
            line["general_kernel_only"] = {"ms_per_step": worst["ms_per_step"], "images_per_sec": worst["images_per_sec"],
                                           "meaning": "same workload, every query block routed to the general (running-max) attention kernel"}
    if cpu:
        # a failure of the host-side numpy / BLAS stack must not take the measured line with it
        try:
            r = subprocess.run([sys.executable, me, "--cpu-baseline-only", "--size", str(args.size), "--quality", str(args.quality),
                                "--arch", args.arch, "--T", str(args.T)], capture_output=True, text=True, timeout=600)
            out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            line["cpu_baseline"] = json.loads(out[-1]) if (r.returncode == 0 and out) else \
                {"value": None, "unit": "images/sec", "cores": 0, "kind": "port",
                 "sample": "cpu baseline process failed (rc %d): %s" % (r.returncode, r.stderr.strip()[-200:])}
        except Exception as e:  # timeout etc.
            line["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port",
                                    "sample": "cpu baseline not measured: %r" % (e,)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--arch", choices=["jpeg", "webp", "avif"], default="jpeg", help="model family (codec inside the loop)")
    ap.add_argument("--T", type=int, default=1000, help="timesteps of the sampling schedule (configs[3]: 250)")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--quality", type=int, default=10)
    ap.add_argument("--attn", choices=["all", "bottleneck"], default="all")
    ap.add_argument("--dtype", choices=["f32", "bf16", "fp8"], default="bf16", help="fp8 = bf16 + e4m3 operands in the deep 3x3 convs (configs[4])")
    ap.add_argument("--codec", choices=["host", "device"], default=None, help="JPEG only: device (default) or host Pillow codec")
    ap.add_argument("--host-codec", choices=["native", "pillow"], default="native", help="WebP / AVIF: library pool or Pillow callback")
    ap.add_argument("--fp8-min-channels", type=int, default=None, choices=[64, 128, 256], help="--dtype fp8: which 3x3 convs take e4m3 operands")
    ap.add_argument("--stream16", type=int, default=None, choices=[-1, 0, 1], help="option bf16_activation_stream (default -1: fp8 mode only)")
    ap.add_argument("--graph", type=int, default=0, help="1 = replay the timestep as captured hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the extra workload slices of the default N=1 run")
    ap.add_argument("--workloads", default=None, help="comma-separated indices into the extra workload list (default: all)")
    ap.add_argument("--full", action="store_true", help="time one complete T-step restoration instead of a slice")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-steps", action="store_true", help=argparse.SUPPRESS)   # child: honour --steps / --batch for the oracle sample
    ap.add_argument("--workloads-child", default=None, help=argparse.SUPPRESS)    # child: run the extra slices, write this file
    return ap.parse_args(argv)


def main():
    args = parse_args()
    T = args.T
    if args.cpu_baseline_only:      # child process of the N=1 run: numpy oracle only, never touches the GPU
        print(json.dumps(cpu_baseline(args.arch, args.size, args.quality, T, n_steps=max(1, args.steps if args.cpu_steps else 1),
                                      batch=args.batch if args.cpu_steps else 1)), flush=True)
        return
    if args.workloads_child:
        workloads_child(args)
        return

    import torch
    from ddpm_image_restoration_amd import dist as D

    env = D.init_from_env()
    if env.world != args.gpus:
        if env.rank == 0:
            print(f"warning: WORLD_SIZE={env.world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", env.local_rank % max(1, torch.cuda.device_count()))   # (modulo: single-GPU rehearsal)
    torch.cuda.set_device(dev)

    B, R, q = args.batch, args.size, args.quality
    head = dict(arch=args.arch, size=R, batch=B, T=T, quality=q, attn=args.attn, dtype=args.dtype, fp8_min_channels=args.fp8_min_channels, stream16=args.stream16)
    K, W = max(1, args.steps), max(0, args.warmup)
    summ, det = run_workload(head, env, dev, K, W, full=args.full, graph=args.graph, host_codec=args.host_codec, codec=args.codec,
                             detail_pass=(args.attn == "bottleneck"))
    is_default = (args.arch, T, R, B, q, args.attn, args.dtype, args.graph, args.codec, args.full) == \
                 ("jpeg", 1000, 256, 16, 10, "all", "bf16", 0, None, False)
    D.barrier(env)
    if env.rank == 0:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(env.world)))
        line = headline_line(args, env.world, env.backend, det, local_world, det.get("codec_threads"))
        if env.world == 1:   # children only at N=1 (every rank would have to take part otherwise), after the headline is measured
            torch.cuda.empty_cache()
            run_children(args, line, extra_workloads=(is_default and not args.no_workloads), cpu=not args.no_cpu_baseline)
        print(fit_line(line), flush=True)
    D.barrier(env)
    if env.world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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

# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (fp8 mode: the roofline kernel -- attention -- stays bf16)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp8": 2500.0}
NOISE_COEFF = {"jpeg": 0.2, "webp": 0.2, "avif": 0.15}
PHASE = {"jpeg": (20, 5, 0.7), "webp": (15, 5, 0.7), "avif": (30, 3, 0.8)}     # quality threshold, period, alpha


def cpu_baseline(arch_name: str, size: int, quality: int, T: int, n_steps: int = 1):
    """numpy oracle ("port") timed on this host's cores on a bounded sample of the same workload:
    B=1 image, `n_steps` timesteps of the T-step schedule at the full image size."""
    import numpy as np
    from ddpm_image_restoration_amd import spec, synth
    from oracle import ddpm_oracle as O
    arch = spec.ARCH_NAMES[arch_name]
    sd = synth.synth_state_dict(arch)
    m = O.Model(arch, sd)
    x0 = synth.synth_images(1, size, seed=123)
    y = O.compress(x0, quality, arch_name)
    x_t = y.copy()
    pq, pevery, palpha = PHASE[arch_name]
    t0 = time.perf_counter()
    for i in range(T - 1, T - 1 - n_steps, -1):
        t = (np.full((1,), i, dtype=np.float32) / np.float32(T)).astype(np.float32)
        th = m(x_t, t, t)
        dec = O.compress(th, quality, arch_name)
        xp = (th - dec) + y
        eps = np.random.default_rng(i).standard_normal(x_t.shape).astype(np.float32)
        x_t = (xp + np.float32(0.85) * (eps * (t * np.float32(NOISE_COEFF[arch_name])).reshape(-1, 1, 1, 1))).astype(np.float32)
        if quality < pq and i % pevery == 0:
            x_t = O.phase_consistency(x_t, y, palpha)
    dt = (time.perf_counter() - t0) / n_steps
    threads = O.oracle_threads()    # the oracle's attention fans out over this many threads (BLAS single-threaded inside)
    return {"value": 1.0 / (dt * T), "unit": "images/sec", "cores": int(threads), "kind": "port",
            "sample": f"numpy oracle, {arch_name} U-Net, B=1 {size}x{size}, {n_steps} of {T} timesteps, {dt:.2f} s/step"}


def committed_traffic(key: str):
    """HBM-side bytes per launch of the roofline kernel from the newest committed PMC summary (profiles/*traffic*.json,
    written by tools/pmc_summary.py from two `rocprofv3 --pmc` passes over this very command: FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE).  rocprofv3 counters cannot be read from inside the run, so the value carries its source."""
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
    return float(e["roofline_kernel_bytes_per_launch"]), {"file": src, "commit": commit, "kernel": e.get("roofline_kernel"),
                                                           "command": e.get("command")}


def main():
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
    ap.add_argument("--graph", type=int, default=0, help="1 = replay the timestep as captured hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full", action="store_true", help="time one complete T-step restoration instead of a slice")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    T = args.T
    if args.cpu_baseline_only:      # child process of the N=1 run: numpy oracle only, never touches the GPU
        print(json.dumps(cpu_baseline(args.arch, args.size, args.quality, T)), flush=True)
        return

    import torch
    from ddpm_image_restoration_amd import dist as D
    from ddpm_image_restoration_amd import harness, spec, synth
    from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
    from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH

    env = D.init_from_env()
    if env.world != args.gpus:
        if env.rank == 0:
            print(f"warning: WORLD_SIZE={env.world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", env.local_rank % max(1, torch.cuda.device_count()))   # (modulo: single-GPU rehearsal)
    torch.cuda.set_device(dev)

    arch = spec.ARCH_NAMES[args.arch]
    B, R, q = args.batch, args.size, args.quality
    model = MODEL_BY_ARCH[arch](device=dev, dtype=args.dtype, attn_levels=args.attn, use_graph=bool(args.graph))
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
    dev_codec = args.arch == "jpeg" and (args.codec or "device") == "device"

    K, W = max(1, args.steps), max(0, args.warmup)
    first = T - 1
    common = dict(steps=T, seed=7, image_offset=lo, device_codec=dev_codec, host_codec=args.host_codec)
    if args.full:
        K, W = 1, 0
    if W > 0:
        sampler.sample(y, q, start_step=first, num_steps=W, **common)
    torch.cuda.synchronize(dev)
    # HIP-event spans force eager launches (events cannot bracket nodes of a replayed graph): with --graph 1 the timed region
    # replays the captured timesteps and the line carries no roofline object
    model.profile(not args.graph)
    D.barrier(env)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if args.full:
        sampler.sample(y, q, **common)
    else:
        sampler.sample(y, q, start_step=first - W, num_steps=K, **common)
    torch.cuda.synchronize(dev)
    D.barrier(env)
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, env, device=dev)
    model.profile(False)

    steps_timed = T if args.full else K
    ms_per_step = dt * 1e3 / steps_timed
    value = n_total / (T * ms_per_step / 1e3)

    a_ms, a_n, a_fl = model.profile_read("attention_l0")
    f_ms, f_n, _ = model.profile_read("forward")
    c_ms, c_n, _ = model.profile_read("codec_host")
    peak = PEAK_TFLOPS[args.dtype]
    heads = 8 if args.arch == "avif" else 4
    roof = None
    if a_n > 0 and a_ms > 0:
        achieved = a_fl / (a_ms * 1e-3) / 1e12
        traffic, tsrc = committed_traffic(f"{args.arch}_{args.dtype}_{R}_B{B}_{args.attn}")
        roof = {"bound": "mfma", "kernel": "flash attention of the full-resolution blocks (down1/up5): N=%d tokens, %d heads, head_dim %d"
                                           % (R * R, heads, 64 // heads),
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": traffic, "traffic_source": tsrc,
                "binding_resource": ("priced against the bf16 MFMA peak as the contract asks; what the kernel saturates is VALU issue "
                                     "(one v_exp_f32 per score: %d FLOP per exp), MFMA pipe ~35-40%% busy -- DESIGN.md 4.1" % (4 * 64 // heads)
                                     if args.dtype == "bf16" else "exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 tiles)"),
                "avg_launch_ms": a_ms / a_n, "launches": a_n, "forward_ms": (f_ms / f_n) if f_n else None}

    if env.rank == 0:
        cfg_no = {("jpeg", 1000): 1, ("webp", 1000): 2, ("avif", 250): 3}.get((args.arch, T))
        is_named = cfg_no is not None and (R, B, q, args.attn, args.dtype) == (256, 16, 10, "all", "bf16")
        codec_txt = ("device JPEG round trip (bit-exact to libjpeg-turbo)" if dev_codec else
                     "host Pillow/libjpeg-turbo callback" if args.arch == "jpeg" else
                     ("library's lib%s C-API thread pool (bit-identical to Pillow), two pipelined micro-batch lanes (full-resolution blocks per lane, the levels below once per timestep)" % args.arch
                      if args.host_codec == "native" else "callback into Pillow (the reference's own calls), two pipelined micro-batch lanes"))
        line = {
            "metric": "restored images/sec (%dx%d, T=%d)" % (R, R, T), "value": value, "unit": "images/sec",
            "n_gpus": env.world, "steps": steps_timed, "warmup": W, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "%s: %s restoration U-Net, %dx%d, batch %d/GPU, T=%d, %s q%d; "
                                   "step = one reverse-diffusion timestep (U-Net forward + codec round trip + DDRM update)"
                                   % (("BASELINE configs[%d]%s" % (cfg_no, "" if cfg_no == 1 else " (per-GPU shape)")) if is_named
                                      else "variant of BASELINE configs", args.arch.upper(), R, R, B, T, args.arch.upper(), q),
                       "global_batch": n_total, "image": R, "T": T, "quality": q, "attn_levels": args.attn,
                       "codec": codec_txt, "timestep_launch": "captured hipGraphs" if args.graph else "eager",
                       "arithmetic": {"bf16": "bf16 matrix-core operands, fp32 accumulation/softmax/GroupNorm statistics, fp32 residual stream",
                                      "fp8": "as bf16, with OCP e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4) in the 3x3 convs with >= 256 channels",
                                      "f32": "fp32 matrix cores"}[args.dtype],
                       "parallelism": f"dp{env.world} (independent image shards, no collective in the loop)"},
            "distributed": {"backend": env.backend if env.world > 1 else None, "world_size": env.world,
                            "weight_blob_bytes": model.weight_blob_bytes(),
                            "load_pack_broadcast_s": load_s},
        }
        if not dev_codec and c_n > 0:
            # host time inside the codec (summed over the lanes of a step) next to the GPU time of the step's forwards: what
            # the two-lane pipeline cannot hide is ms_per_step - forwards - the step's small device kernels
            fwd_per_step = f_ms / steps_timed if f_n else None
            line["codec"] = {"host_ms_per_step": c_ms / steps_timed, "calls_per_step": c_n / steps_timed,
                             "forward_ms_per_step": fwd_per_step,
                             "not_hidden_ms_per_step": (max(0.0, ms_per_step - fwd_per_step) if fwd_per_step else None)}
        if roof:
            line["roofline"] = roof
        if not args.no_cpu_baseline and env.world == 1:     # the CPU baseline is timed at N=1 only
            # in a child process: a failure of the host-side numpy / BLAS stack must not take the measured line with it
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(R),
                                    "--quality", str(q), "--arch", args.arch, "--T", str(T)], capture_output=True, text=True, timeout=600)
                out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                line["cpu_baseline"] = json.loads(out[-1]) if (r.returncode == 0 and out) else \
                    {"value": None, "unit": "images/sec", "cores": 0, "kind": "port",
                     "sample": "cpu baseline process failed (rc %d): %s" % (r.returncode, r.stderr.strip()[-200:])}
            except Exception as e:  # timeout etc.
                line["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port",
                                        "sample": "cpu baseline not measured: %r" % (e,)}
        print(json.dumps(line), flush=True)
    D.barrier(env)
    if env.world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

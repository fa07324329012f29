#!/usr/bin/env python3
"""Headline benchmark: restored images/sec of the DDRM reverse-diffusion sampler on MI355X.

Workload (BASELINE.json configs[1]): JPEG restoration U-Net (reference architecture: self-attention over all
H*W tokens in every one of the 13 blocks), 256x256, batch 16 per GPU, T=1000, JPEG q10, synthetic
JPEG-degraded inputs and synthetic weights (the reference ships no checkpoint).

A "step" is ONE reverse-diffusion timestep over the batch: U-Net forward + uint8 quantise + JPEG round trip +
DDRM update (+ the FFT phase fix on every 5th index, as q10 < 20) -- webp_inference.py:566-600.  All T
timesteps do identical work, so  images/sec = N_gpus * B / (T * seconds_per_step); the K timed steps are
consecutive indices of the real T=1000 schedule run through the product C ABI (ddpmir_ddrm_sample_loop with
start_step/num_steps).  `--full` runs the whole T-step restoration instead (minutes).

  python bench.py                       # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
T_STEPS = 1000


def cpu_baseline(size: int, quality: int, n_steps: int = 1):
    """numpy oracle ("port") timed on this host's cores on a bounded sample of the same workload:
    B=1 image, `n_steps` timesteps of the T=1000 schedule at the full 256x256 size."""
    import numpy as np
    from ddpm_image_restoration_amd import spec, synth
    from oracle import ddpm_oracle as O
    sd = synth.synth_state_dict(spec.ARCH_JPEG)
    m = O.Model(spec.ARCH_JPEG, sd)
    x0 = synth.synth_images(1, size, seed=123)
    y = O.compress(x0, quality, "jpeg")
    x_t = y.copy()
    t0 = time.perf_counter()
    for i in range(T_STEPS - 1, T_STEPS - 1 - n_steps, -1):
        t = (np.full((1,), i, dtype=np.float32) / np.float32(T_STEPS)).astype(np.float32)
        th = m(x_t, t, t)
        dec = O.compress(th, quality, "jpeg")
        xp = (th - dec) + y
        eps = np.random.default_rng(i).standard_normal(x_t.shape).astype(np.float32)
        x_t = (xp + np.float32(0.85) * (eps * (t * np.float32(0.2)).reshape(-1, 1, 1, 1))).astype(np.float32)
        if quality < 20 and i % 5 == 0:
            x_t = O.phase_consistency(x_t, y, 0.7)
    dt = (time.perf_counter() - t0) / n_steps
    threads = O.oracle_threads()    # the oracle's attention fans out over this many threads (BLAS single-threaded inside)
    return {"value": 1.0 / (dt * T_STEPS), "unit": "images/sec", "cores": int(threads), "kind": "port",
            "sample": f"numpy oracle, B=1 {size}x{size}, {n_steps} of {T_STEPS} timesteps, {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--quality", type=int, default=10)
    ap.add_argument("--attn", choices=["all", "bottleneck"], default="all")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16")
    ap.add_argument("--codec", choices=["host", "device"], default="device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full", action="store_true", help="time one complete T-step restoration instead of a slice")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:      # child process of the N=1 run: numpy oracle only, never touches the GPU
        print(json.dumps(cpu_baseline(args.size, args.quality)), flush=True)
        return

    import numpy as np
    import torch
    from ddpm_image_restoration_amd import dist as D
    from ddpm_image_restoration_amd import codec, spec, synth
    from ddpm_image_restoration_amd.model import JPEGDiffusionModel
    from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler

    env = D.init_from_env()
    if env.world != args.gpus:
        if env.rank == 0:
            print(f"warning: WORLD_SIZE={env.world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", env.local_rank % max(1, torch.cuda.device_count()))   # (modulo: single-GPU rehearsal)
    torch.cuda.set_device(dev)

    B, R, q = args.batch, args.size, args.quality
    model = JPEGDiffusionModel(device=dev, dtype=args.dtype, attn_levels=args.attn)
    D.load_weights_broadcast(model, lambda: synth.synth_state_dict(spec.ARCH_JPEG), env)
    sampler = DDRMJPEGSampler(model)

    # synthetic JPEG-degraded batch of this rank's shard (global image index = rank*B + j)
    n_total = B * env.world
    lo, hi = D.shard_range(n_total, env.rank, env.world)
    x0 = synth.synth_images(n_total, R, seed=123)[lo:hi]
    y = codec.jpeg_compress(torch.from_numpy(x0), q).to(dev)     # y = JPEG(x0, q): the product's own host codec
    dev_codec = args.codec == "device"

    K, W = max(1, args.steps), max(0, args.warmup)
    first = T_STEPS - 1
    common = dict(steps=T_STEPS, seed=7, image_offset=lo, device_codec=dev_codec)
    if args.full:
        K, W = 1, 0
    if W > 0:
        sampler.sample(y, q, start_step=first, num_steps=W, **common)
    torch.cuda.synchronize(dev)
    model.profile(True)
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

    steps_timed = T_STEPS if args.full else K
    ms_per_step = dt * 1e3 / steps_timed
    value = n_total / (T_STEPS * ms_per_step / 1e3)

    a_ms, a_n, a_fl = model.profile_read("attention_l0")
    f_ms, f_n, _ = model.profile_read("forward")
    peak = PEAK_TFLOPS[args.dtype]
    roof = None
    if a_n > 0 and a_ms > 0:
        achieved = a_fl / (a_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "flash attention, N=%d tokens, head_dim 16 (down1/up5)" % (R * R),
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                # HBM-side bytes per launch from the committed PMC passes over this very command (tools/pmc_traffic.sh ->
                # profiles/r01_bench_bf16_pmc_{fetch,write}.csv, level-0 launches at B=16: FETCH_SIZE 1 147 940 KiB x2 [gfx950
                # correction] + WRITE_SIZE 131 072 KiB), scaled by B/16; rocprofv3 counters cannot be read from inside the run
                "traffic": (2 * 1147939.6 + 131072.0) * 1024 * (B / 16.0) if (args.dtype == "bf16" and R == 256) else None,
                "valu_bound_note": ("bound by v_exp_f32 issue (32 FLOP per exp at head_dim 16): VALU pipe ~80% busy, MFMA pipe ~39%"
                                    if args.dtype == "bf16" else
                                    "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); the 32-row PV tile is half empty at head_dim 16"),
                "avg_launch_ms": a_ms / a_n, "launches": a_n, "forward_ms": (f_ms / f_n) if f_n else None}

    if env.rank == 0:
        line = {
            "metric": "restored images/sec (%dx%d, T=%d)" % (R, R, T_STEPS), "value": value, "unit": "images/sec",
            "n_gpus": env.world, "steps": steps_timed, "warmup": W, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "%s: JPEG restoration U-Net, %dx%d, batch %d/GPU, T=%d, JPEG q%d; "
                                   "step = one reverse-diffusion timestep (U-Net forward + codec round trip + DDRM update)"
                                   % ("BASELINE configs[1]" if (R, B, q, args.attn, args.dtype) == (256, 16, 10, "all", "bf16")
                                      else "variant of BASELINE configs[1]", R, R, B, T_STEPS, q),
                       "global_batch": n_total, "image": R, "T": T_STEPS, "quality": q, "attn_levels": args.attn,
                       "codec": "device JPEG round trip (bit-exact to libjpeg-turbo)" if dev_codec else "host Pillow/libjpeg-turbo callback",
                       "arithmetic": "bf16 matrix-core operands, fp32 accumulation/softmax/GroupNorm statistics, fp32 activations in HBM" if args.dtype == "bf16" else "fp32 matrix cores",
                       "parallelism": f"dp{env.world} (independent image shards, no collective in the loop)"},
        }
        if roof:
            line["roofline"] = roof
        if not args.no_cpu_baseline and env.world == 1:     # the CPU baseline is timed at N=1 only
            # in a child process: a failure of the host-side numpy / BLAS stack must not take the measured line with it
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(R),
                                    "--quality", str(q)], capture_output=True, text=True, timeout=600)
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

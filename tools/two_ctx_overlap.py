"""Does running two half-batches CONCURRENTLY (two contexts = two streams, two host threads) beat one full batch?
VERDICT r3 item 5: the level-0 attention is VALU-issue-bound with the matrix pipe ~35 % busy and HBM idle, the deep levels are
latency-bound with few workgroups -- if the hardware co-schedules one lane's GEMM / streaming kernels under the other lane's
attention, two offset lanes would hide part of the ~12 ms outside the level-0 attention.  Zero-code experiment first."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

dev = torch.device("cuda", 0)
sd = synth.synth_state_dict(spec.ARCH_JPEG)


def run(s, y, q, k, first):
    s.sample(y, q, steps=1000, start_step=first, num_steps=k, seed=1)


for name, kw, R, B, q, K in (("headline 256x256 all-attention", dict(dtype="bf16"), 256, 16, 10, 6),
                             ("configs[4] 512x512 bottleneck", dict(dtype="bf16", attn_levels="bottleneck"), 512, 8, 20, 20)):
    ms = [JPEGDiffusionModel(device=dev, **kw) for _ in range(2)]
    for m in ms:
        m.load_state_dict(sd)
    y = codec.jpeg_compress(torch.from_numpy(synth.synth_images(B, R, seed=3)), q).to(dev)
    ss = [DDRMJPEGSampler(m) for m in ms]
    h = B // 2
    ys = [y[:h].contiguous(), y[h:].contiguous()]
    for rep in range(2):
        run(ss[0], y, q, 2, 999)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(ss[0], y, q, K, 997)
        torch.cuda.synchronize()
        full = (time.perf_counter() - t0) / K * 1e3
        run(ss[0], ys[0], q, 2, 999)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(ss[0], ys[0], q, K, 997)
        torch.cuda.synchronize()
        half = (time.perf_counter() - t0) / K * 1e3
        for i in range(2):
            run(ss[i], ys[i], q, 2, 999)
        torch.cuda.synchronize()
        # lane 1 starts half a timestep late so that its deep levels meet lane 0's full-resolution work
        def worker(i, delay):
            time.sleep(delay)
            run(ss[i], ys[i], q, K, 997)
        for delay in (0.0, half * 0.5e-3):
            th = [threading.Thread(target=worker, args=(i, delay * i)) for i in range(2)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            both = (time.perf_counter() - t0 - delay) / K * 1e3
            print(f"{name}: one context B={B}: {full:.2f} ms/step | one context B={h}: {half:.2f} | two contexts B={h} each, concurrent"
                  f" (offset {delay * 1e3:.1f} ms): {both:.2f} ms/step for all {B} images  -> x{full / both:.3f}", flush=True)
    del ss, ms
    torch.cuda.empty_cache()

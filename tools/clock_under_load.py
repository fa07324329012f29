"""the shader clock while the library's own kernels run.  A one-wave sampler kernel on its own stream
(ops.clock_sample_*) compares s_memtime with s_memrealtime over a window inside a longer run of
  (a) nothing (idle device),  (b) the headline workload (79 % level-0 attention),  (c) the same with every query block on the general
  attention kernel,  (d) configs[4]'s shape in bf16 and fp8 mode (GEMM / streaming passes, attention at the bottleneck only)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, ops, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

dev = torch.device("cuda", 0)
sd = synth.synth_state_dict(spec.ARCH_JPEG)
ops.clock_sample_begin(200)
time.sleep(0.25)
print("idle device: %.3f GHz over %.0f ms" % ops.clock_sample_end(), flush=True)
for name, kw, R, B, q, K, nm in (("headline (256x256, B=16, attention everywhere)", dict(dtype="bf16"), 256, 16, 10, 24, None),
                                 ("headline, general attention kernel only", dict(dtype="bf16"), 256, 16, 10, 24, -1.0),
                                 ("configs[4] shape bf16", dict(dtype="bf16", attn_levels="bottleneck"), 512, 8, 20, 120, None),
                                 ("configs[4] shape fp8", dict(dtype="fp8", attn_levels="bottleneck"), 512, 8, 20, 120, None)):
    m = JPEGDiffusionModel(device=dev, **kw)
    if nm is not None:
        m.set_option("attention_no_max_bound", nm)
    m.load_state_dict(sd)
    y = codec.jpeg_compress(torch.from_numpy(synth.synth_images(B, R, seed=3)), q).to(dev)
    s = DDRMJPEGSampler(m)
    s.sample(y, q, steps=1000, start_step=999, num_steps=3, seed=1)          # warm (plans, first-touch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = threading.Thread(target=lambda: s.sample(y, q, steps=1000, start_step=996, num_steps=K, seed=1))
    th.start()
    time.sleep(0.35)                                                          # let the run reach its steady power state
    ops.clock_sample_begin(500)
    ghz, ms = ops.clock_sample_end()
    th.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {ghz:.3f} GHz over {ms:.0f} ms inside a {dt * 1e3:.0f} ms run ({dt / K * 1e3:.2f} ms/step)", flush=True)
    del s, m
    torch.cuda.empty_cache()

"""Yardstick, not product: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the plain-GEMM equivalents of the
3x3 conv shapes (M = B*H*W pixels, K = 9*Cin, N = Cout, bf16 in / bf16 out), beside gemm16's time for the real conv.
usage: python tools/gemm_yardstick.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpm_image_restoration_amd import ops

shapes = [(16, 256, 64, 64), (16, 128, 128, 128), (16, 64, 256, 256), (16, 32, 512, 512), (16, 64, 512, 128), (16, 128, 256, 64),
          (8, 512, 64, 64), (8, 256, 128, 128), (8, 128, 256, 256), (8, 64, 512, 512)]
for B, h, ci, co in shapes:
    M, K, N = B * h * h, 9 * ci, co
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.matmul(a, w)
    e1.record(); torch.cuda.synchronize()
    ms_lt = e0.elapsed_time(e1) / 10
    del a, w
    ms_conv = min(ops.conv_time(B, h, h, ci, co, 3, "dma", True, False, 0, 5) for _ in range(2))
    fl = 2.0 * M * K * N
    print(f"B={B:2d} H={h:3d} {ci:4d}->{co:4d} k3  M={M:7d} K={K:5d} N={N:4d}: vendor plain GEMM {ms_lt*1e3:7.1f} us {fl/ms_lt/1e9:7.1f} TF/s | "
          f"gemm16 conv {ms_conv*1e3:7.1f} us {fl/ms_conv/1e9:7.1f} TF/s", flush=True)

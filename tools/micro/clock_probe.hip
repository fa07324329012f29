// what is the shader clock under the attention kernel's instruction mix?  The level-0 attention launch was
// measured at 2.13 GHz effective (nominal 2.4).  This probe runs, register-only (no LDS, no memory traffic), for ~10 ms each:
//   mix      2 x [1 QK MFMA, 16 v_exp_f32, 8 v_cvt_pk_bf16_f32, 2 PV MFMA]   (the d = 16 inner loop)
//   exp      the same without the MFMAs          mfma   the same without exp / cvt          fma   plain v_fma_f32 only
// at 1, 2 and 3 waves per SIMD on every CU, and reads s_memtime (shader clock) against s_memrealtime (100 MHz) in workgroup 0:
// effective GHz = shader cycles / wall time.  If the clock drops with the MFMA share, the attention kernel's 199 "cycles" per
// 1024 scores (computed at a nominal clock) overstate its issue-slot use and the power cap, not issue, is the binder.
//   build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>   // 0 mix, 1 exp + cvt only, 2 MFMA only, 3 FMA only
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed, unsigned long long* clk) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  bf16x8 k, v0, v1, qq[2];
  for (int j = 0; j < 8; ++j) { k[j] = (__bf16)(seed + j * 0.01f); v0[j] = (__bf16)(0.5f + j * 0.01f); v1[j] = (__bf16)(0.25f); qq[0][j] = (__bf16)(0.01f * (threadIdx.x & 7)); qq[1][j] = (__bf16)(0.02f * (threadIdx.x & 7) + 1.f); }
  f32x16 o, nm[2], s;
  for (int r = 0; r < 16; ++r) { o[r] = 0.f; nm[0][r] = -3.0f; nm[1][r] = -4.0f; s[r] = -0.001f * (threadIdx.x + r + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (MODE == 0 || MODE == 2) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, qq[u], nm[u], 0, 0, 0);
      bf16x8 pb[2];
      if (MODE == 0 || MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pb[r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(s[r]);
      } else if (MODE == 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_fmaf(s[r], 0.999f, 0.001f);
        pb[0] = v0; pb[1] = v1;
      } else { pb[0] = v0; pb[1] = v1; }
      if (MODE == 0 || MODE == 2) {
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb[0], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb[1], o, 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] += (float)pb[0][j]; o[8 + j] += (float)pb[1][j]; }
      }
      if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s[r] * 0.999f;
      }
    }
    k[0] = (__bf16)((float)k[0] * 0.999f);
  }
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += o[r] + s[r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  unsigned long long* clk; hipMalloc(&clk, 16);
  const char* names[4] = {"mix (exp + cvt + 3 MFMA)", "exp + cvt only", "MFMA only", "v_fma_f32 only"};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 4; ++mode)
    for (int w = 1; w <= 3; ++w) {
      const int grid = 256 * w, iters = mode == 2 ? 120000 / w : mode == 3 ? 150000 / w : 40000 / w;
      auto launch = [&](int it) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, out, it, 0.1f, clk); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, out, it, 0.1f, clk); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, out, it, 0.1f, clk); break;
          default: hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, 0, out, it, 0.1f, clk); break;
        }
      };
      for (int rep = 0; rep < 24; ++rep) launch(iters);     // ~0.2-0.3 s of this mix: reach its steady power state
      hipDeviceSynchronize();
      hipEventRecord(a); launch(iters); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double ghz = (double)h[0] / ((double)h[1] * 10.0);        // wall ticks are 10 ns
      const double tiles_per_simd = (double)grid * 4 * iters * 2 / 1024.0;
      printf("%-26s waves/SIMD=%d: %7.2f ms, shader clock %.3f GHz; %6.1f ns = %5.0f shader cycles per 1024-score tile per SIMD\n", names[mode], w, ms, ghz,
             ms * 1e6 / tiles_per_simd, ms * 1e6 / tiles_per_simd * ghz);
    }
  return 0;
}

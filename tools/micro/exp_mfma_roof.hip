// Microbenchmark: the instruction mix of the d=16 flash-attention inner loop with all operands in registers
// (no LDS, no barriers): 2 x [1 QK MFMA, 16 v_exp, 8 v_cvt_pk, 2 PV MFMA] per iteration.  Gives the VALU-issue
// ceiling of this mix on the chip (cycles per 1024 scores per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int QS>
__global__ __launch_bounds__(256) void mix(float* out, int iters, float seed) {
  bf16x8 k, v0, v1, q[QS];
  for (int j = 0; j < 8; ++j) { k[j] = (__bf16)(seed + j * 0.01f); v0[j] = (__bf16)(0.5f + j * 0.01f); v1[j] = (__bf16)(0.25f); }
  f32x16 o[QS], negm[QS];
  for (int u = 0; u < QS; ++u) {
    for (int j = 0; j < 8; ++j) q[u][j] = (__bf16)(0.01f * (threadIdx.x & 7) + u);
    for (int r = 0; r < 16; ++r) { o[u][r] = 0.f; negm[u][r] = -3.0f - u; }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, q[u], negm[u], 0, 0, 0);
      bf16x8 pb[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) pb[r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(s[r]);
      o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb[0], o[u], 0, 0, 0);
      o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb[1], o[u], 0, 0, 0);
    }
    k[0] = (__bf16)((float)k[0] * 0.999f);   // keep the loop from being hoisted
  }
  float acc = 0.f;
  for (int u = 0; u < QS; ++u) for (int r = 0; r < 16; ++r) acc += o[u][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4096;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
    for (int qs = 1; qs <= 2; ++qs) {
      int grid = 256 * wgs_per_cu;
      auto launch = [&] { if (qs == 1) hipLaunchKernelGGL(mix<1>, dim3(grid), dim3(256), 0, 0, out, iters, 0.1f); else hipLaunchKernelGGL(mix<2>, dim3(grid), dim3(256), 0, 0, out, iters, 0.1f); };
      launch(); hipDeviceSynchronize();
      hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double tiles = (double)grid * 4 * iters * qs;            // wave-tiles of 1024 scores
      double per_simd = tiles / 1024.0;                        // tiles per SIMD
      printf("waves/SIMD=%d QS=%d: %.3f ms, %.1f ns per 1024 scores per SIMD (= %.0f cycles at 2.1 GHz), %.2f Tscores/s\n",
             wgs_per_cu, qs, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.1, tiles * 1024 / ms / 1e9);
    }
  }
  return 0;
}

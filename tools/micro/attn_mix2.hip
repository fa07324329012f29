// Microbenchmark: which VALU instructions of the head_dim-16 flash-attention tile loop cost what, registers only.
// Per tile and query sub-tile: 1 QK^T MFMA, 16 exps, the bf16 packing of p, 2 PV MFMAs.  Variants:
//   0 cvt      p packed with v_cvt_pk_bf16_f32 (round to nearest even)                      -- the round-1 kernel
//   1 perm     p packed with v_perm_b32 (upper halves of two fp32 registers: truncation)
//   2 expOnly  exps, no packing (PV consumes stale operands): cost of the transcendental stream alone
//   3 cvtOnly  packing with v_cvt_pk, no exps
//   4 permOnly packing with v_perm, no exps
//   5 mfmaOnly the three MFMAs alone
// Prints cycles per 1024 scores per SIMD at 1, 2, 3 waves per SIMD (QS = 2 sub-tiles per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int VAR, int QS>
__global__ __launch_bounds__(256) void mix(float* out, int iters, float seed) {
  bf16x8 k, v0, v1, q[QS];
  for (int j = 0; j < 8; ++j) { k[j] = (__bf16)(seed + j * 0.01f); v0[j] = (__bf16)(0.5f + j * 0.01f); v1[j] = (__bf16)(0.25f); }
  f32x16 o[QS], negm[QS];
  bf16x8 pb[QS][2];
  for (int u = 0; u < QS; ++u) {
    for (int j = 0; j < 8; ++j) { q[u][j] = (__bf16)(0.01f * (threadIdx.x & 7) + u); pb[u][0][j] = (__bf16)0.5f; pb[u][1][j] = (__bf16)0.25f; }
    for (int r = 0; r < 16; ++r) { o[u][r] = 0.f; negm[u][r] = -3.0f - u; }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, q[u], negm[u], 0, 0, 0);
      if (VAR == 0 || VAR == 1 || VAR == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
      }
      if (VAR == 0 || VAR == 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pb[u][r >> 3][r & 7] = (__bf16)s[r];
      } else if (VAR == 1 || VAR == 4) {
        u32x4 w[2];
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          w[r >> 3][(r & 7) >> 1] = __builtin_amdgcn_perm(__float_as_uint(s[r + 1]), __float_as_uint(s[r]), 0x07060302u);
        pb[u][0] = __builtin_bit_cast(bf16x8, w[0]);
        pb[u][1] = __builtin_bit_cast(bf16x8, w[1]);
      } else if (VAR == 2) {
        // keep the exps alive without packing: fold them into the accumulator init of the next tile
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[u][r] = s[r] * 0.f - 3.0f;
      }
      o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb[u][0], o[u], 0, 0, 0);
      o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb[u][1], o[u], 0, 0, 0);
    }
    k[0] = (__bf16)((float)k[0] * 0.999f);   // keep the loop from being hoisted
  }
  float acc = 0.f;
  for (int u = 0; u < QS; ++u) for (int r = 0; r < 16; ++r) acc += o[u][r] + negm[u][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int VAR>
void run(const char* name, float* out) {
  const int iters = 4096;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("%-9s", name);
  for (int wps = 1; wps <= 3; ++wps) {
    const int grid = 256 * wps;
    hipLaunchKernelGGL((mix<VAR, 2>), dim3(grid), dim3(256), 0, 0, out, iters, 0.1f); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL((mix<VAR, 2>), dim3(grid), dim3(256), 0, 0, out, iters, 0.1f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double tiles = (double)grid * 4 * iters * 2, per_simd = tiles / 1024.0;
    printf("  %dw/SIMD: %6.1f ns = %4.0f cyc @2.1GHz (%5.2f Tscores/s)", wps, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.1, tiles * 1024 / ms / 1e9);
  }
  printf("\n");
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  run<0>("cvt", out); run<1>("perm", out); run<2>("expOnly", out); run<3>("cvtOnly", out); run<4>("permOnly", out); run<5>("mfmaOnly", out);
  return 0;
}

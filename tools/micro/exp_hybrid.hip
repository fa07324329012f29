// Can part of the softmax exponentials run on the plain VALU in the shadow of the transcendental unit?  tools/micro/exp_rate.hip:
// a v_exp_f32 keeps the transcendental pipe busy for ~9.3 (nominal) cycles per wave instruction while a v_cvt_pk beside it costs
// almost nothing -- the VALU issue port is free most of that time.  This probe runs the d = 16 attention inner-loop mix of
// exp_mfma_roof.hip (per 32 x 32 score tile: 1 QK MFMA, 16 exponentials, 8 v_cvt_pk_bf16_f32, 2 PV MFMAs; two tiles per iteration)
// with NP of the 16 exponentials of every tile evaluated as  2^x = ldexp(poly3(x - rint(x)), rint(x))  in 7 plain VALU
// instructions (magic-number rounding, three FMAs, exponent insertion by v_lshl_add_u32) instead of v_exp_f32.
//   build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form exp_hybrid.hip -o exp_hybrid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float exp2_poly(float x) {
  const float t = x + 12582912.0f;                       // 1.5 * 2^23: the sum's low mantissa bits hold rint(x)
  const float f = x - (t - 12582912.0f);                 // in [-0.5, 0.5]
  // 2^f on [-0.5, 0.5], degree 3 (max relative error 1.1e-4: a 28th of a bf16 rounding step)
  float p = fmaf(f, 0.0555041086f, 0.2402264923f);
  p = fmaf(f, p, 0.6931471806f);
  p = fmaf(f, p, 1.0f);
  return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t) << 23));
}

template <int NP>
__global__ __launch_bounds__(256) void mix(float* out, int iters, float seed) {
  bf16x8 k, v0, v1, q[2];
  for (int j = 0; j < 8; ++j) { k[j] = (__bf16)(seed + j * 0.01f); v0[j] = (__bf16)(0.5f + j * 0.01f); v1[j] = (__bf16)(0.25f); }
  f32x16 o[2], negm[2];
  for (int u = 0; u < 2; ++u) {
    for (int j = 0; j < 8; ++j) q[u][j] = (__bf16)(0.01f * (threadIdx.x & 7) + u);
    for (int r = 0; r < 16; ++r) { o[u][r] = 0.f; negm[u][r] = -3.0f - u; }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, q[u], negm[u], 0, 0, 0);
      bf16x8 pb[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // spread the polynomial evaluations evenly between the v_exp_f32 instructions
        const bool poly = NP > 0 && ((r * NP) / 16 != ((r + 1) * NP) / 16);
        pb[r >> 3][r & 7] = (__bf16)(poly ? exp2_poly(s[r]) : __builtin_amdgcn_exp2f(s[r]));
      }
      o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb[0], o[u], 0, 0, 0);
      o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb[1], o[u], 0, 0, 0);
    }
    k[0] = (__bf16)((float)k[0] * 0.999f);
  }
  float acc = 0.f;
  for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) acc += o[u][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void check(float* err) {   // max relative error of the polynomial against exp2f over the score range
  float worst = 0.f;
  for (int i = threadIdx.x; i < 400000; i += blockDim.x) {
    const float x = -90.f + i * (170.f / 400000.f);
    const float a = exp2_poly(x), b = exp2f(x);
    worst = fmaxf(worst, fabsf(a - b) / b);
  }
  atomicMax(reinterpret_cast<unsigned*>(err), __float_as_uint(worst));
}

template <int NP>
void run(float* out, int w) {
  const int iters = 4096, grid = 256 * w;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(mix<NP>, dim3(grid), dim3(256), 0, 0, out, iters, 0.1f); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(mix<NP>, dim3(grid), dim3(256), 0, 0, out, iters, 0.1f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double per_simd = (double)grid * 4 * iters * 2 / 1024.0;
  printf("waves/SIMD=%d, %2d of 16 exponentials on the plain VALU: %7.3f ms = %6.1f ns per 1024-score tile per SIMD\n", w, NP, ms, ms * 1e6 / per_simd);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  float* err; hipMalloc(&err, 4); hipMemset(err, 0, 4);
  hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, 0, err);
  float e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
  printf("exp2_poly: max relative error %.3g over [-90, 80]\n", e);
  for (int w = 2; w <= 3; ++w) {
    run<0>(out, w); run<1>(out, w); run<2>(out, w); run<3>(out, w); run<4>(out, w); run<6>(out, w); run<8>(out, w);
  }
  return 0;
}

// Issue rate of v_exp_f32 vs v_exp_f16 (+ the conversions a f16 softmax would need) on gfx950, one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = -0.001f * (threadIdx.x + i + 1);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(v[i]));
    } else if (MODE == 5) {   // reference points: a plain full-rate VALU op and a cvt
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(v[i]));
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[i + 1]));
    } else if (MODE == 2) {   // f16 path: cvt_pkrtz (1 per 2) + exp_f16 low + exp_f16 high (SDWA) per pair
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[i + 1]));
        asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(v[i]));
        asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(v[i]));
      }
    } else {                  // today's path: exp_f32 x2 + cvt_pk_bf16 per pair
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[i + 1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[i + 1]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wgs_per_cu) {
  float* out; hipMalloc(&out, 256 * 1024 * 256 * 4);
  const int iters = 20000, grid = 256 * wgs_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per SIMD: wgs_per_cu waves, each iters*16 element-ops
  double cyc = ms * 1e-3 * 2.1e9 / ((double)iters * 16 * wgs_per_cu);
  printf("%-44s waves/SIMD=%d: %.2f cycles per element-op at 2.1 GHz (%.3f ms)\n", name, wgs_per_cu, cyc, ms);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<0>("v_exp_f32", w);
    run<1>("v_exp_f16", w);
    run<2>("cvt_pkrtz + 2x v_exp_f16_sdwa per pair", w);
    run<3>("2x v_exp_f32 + cvt_pk_bf16 per pair (today)", w);
    run<4>("v_exp_legacy_f32", w);
    run<5>("v_mul_f32 (full-rate reference)", w);
    run<6>("v_cvt_pk_bf16_f32 (8 per 16 elements)", w);
  }
  return 0;
}

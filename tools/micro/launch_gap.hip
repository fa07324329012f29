// What puts 5-6 us between some back-to-back dispatches of one stream (profiles/r03_c5_bf16_forward_timeline.txt: 0 us between
// pointwise kernels, 5.6 us before and after every kernel with dynamic LDS)?  Each variant launches N kernels that spin for a
// fixed number of clocks (so the host runs ahead) and reports (elapsed / N - spin time).
// build: hipcc -O3 --offload-arch=gfx950 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { long long v[60]; };   // 480-byte kernel argument, like IGemm

__device__ __forceinline__ void spin(long long clocks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(2);
}
__global__ void k_plain(float* o, long long c) { spin(c); if (c < 0) o[threadIdx.x] = 1.f; }
__global__ void k_dyn(float* o, long long c) {
  extern __shared__ float sm[];
  spin(c);
  if (c < 0) { sm[threadIdx.x] = 1.f; o[threadIdx.x] = sm[threadIdx.x ^ 1]; }
}
__global__ void k_static(float* o, long long c) {
  __shared__ float sm[12 * 1024];
  spin(c);
  if (c < 0) { sm[threadIdx.x] = 1.f; o[threadIdx.x] = sm[threadIdx.x ^ 1]; }
}
__global__ void k_big(float* o, long long c, Big b) { spin(c); if (c < 0) o[threadIdx.x] = (float)b.v[threadIdx.x % 60]; }
__global__ void __launch_bounds__(256, 1) k_vgpr(float* o, long long c) {
  spin(c);
  asm volatile("v_mov_b32 v250, 0" ::: "v250");
  if (c < 0) o[threadIdx.x] = 1.f;
}

int main() {
  float* o; hipMalloc(&o, 4096);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int N = 400; const long long C = 2000;   // wall_clock64 ticks at 100 MHz: 20 us
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dyn), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  Big big{};
  auto run = [&](const char* what, auto&& body) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a, s);
      for (int i = 0; i < N; ++i) body(i);
      hipEventRecord(b, s);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("%-64s %7.2f us per launch beyond the 20 us spin\n", what, ms * 1e3 / N - 20.0);
    }
  };
  const dim3 g(256), t(256);   // one workgroup per CU: every variant is resident at once
  run("plain, no LDS", [&](int) { hipLaunchKernelGGL(k_plain, g, t, 0, s, o, C); });
  run("dynamic LDS 1 KiB", [&](int) { hipLaunchKernelGGL(k_dyn, g, t, 1024, s, o, C); });
  run("dynamic LDS 48 KiB", [&](int) { hipLaunchKernelGGL(k_dyn, g, t, 48 * 1024, s, o, C); });
  run("dynamic LDS 80 KiB (needs the attribute)", [&](int) { hipLaunchKernelGGL(k_dyn, g, t, 80 * 1024, s, o, C); });
  run("static LDS 48 KiB", [&](int) { hipLaunchKernelGGL(k_static, g, t, 0, s, o, C); });
  run("480-byte argument", [&](int) { hipLaunchKernelGGL(k_big, g, t, 0, s, o, C, big); });
  run("256-VGPR kernel", [&](int) { hipLaunchKernelGGL(k_vgpr, g, t, 0, s, o, C); });
  run("alternating plain / dynamic 48 KiB", [&](int i) {
    if (i & 1) hipLaunchKernelGGL(k_dyn, g, t, 48 * 1024, s, o, C); else hipLaunchKernelGGL(k_plain, g, t, 0, s, o, C); });
  run("alternating plain / static 48 KiB", [&](int i) {
    if (i & 1) hipLaunchKernelGGL(k_static, g, t, 0, s, o, C); else hipLaunchKernelGGL(k_plain, g, t, 0, s, o, C); });
  run("alternating plain / 480-byte argument", [&](int i) {
    if (i & 1) hipLaunchKernelGGL(k_big, g, t, 0, s, o, C, big); else hipLaunchKernelGGL(k_plain, g, t, 0, s, o, C); });
  run("alternating dynamic 16 KiB / dynamic 48 KiB", [&](int i) {
    hipLaunchKernelGGL(k_dyn, g, t, (i & 1) ? 48 * 1024 : 16 * 1024, s, o, C); });
  return 0;
}

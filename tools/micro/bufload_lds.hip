// Semantics of buffer_load_dwordx4 ... lds (LDS-DMA through a raw buffer descriptor) on gfx950, as the conv kernels rely on:
//  (1) in-range lanes copy 16 bytes to LDS at M0 base + lane * 16;  (2) a lane whose voffset is out of range gets ZEROS;
//  (3) how soffset takes part in the range check.   build: hipcc -O3 --offload-arch=gfx950 bufload_lds.hip -o bufload_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* in, float* out, unsigned bytes, unsigned soff, unsigned bad_lane, unsigned bad_voff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 64 * 4; i += 64) reinterpret_cast<float*>(smem)[i] = -7.0f;   // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)bytes, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned voff = threadIdx.x * 16u;
  if (threadIdx.x == bad_lane) voff = bad_voff;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(size_t)lds0, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
  const int N = 4096;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<float> r(256);
  auto run = [&](const char* what, unsigned bytes, unsigned soff, unsigned bad_lane, unsigned bad_voff) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, o, bytes, soff, bad_lane, bad_voff);
    hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    printf("%-62s lane0 %.0f lane1 %.0f lane3 %.0f %.0f %.0f %.0f lane63 %.0f\n", what, r[0], r[4], r[12], r[13], r[14], r[15], r[252]);
  };
  run("in range, soffset 0: expect 0 4 12..15 252", N * 4, 0, 999, 0);
  run("lane 3 voffset 0xFFFFFFF0 (out of range): expect zeros", N * 4, 0, 3, 0xFFFFFFF0u);
  run("lane 3 voffset = num_records: expect zeros", N * 4, 0, 3, N * 4);
  run("soffset 4096 bytes: expect 1024 1028 1036.. 1276", N * 4, 4096, 999, 0);
  run("num_records 1024 B, soffset 4096 (voffset in range): zeros iff soffset is range-checked", 1024, 4096, 999, 0);
  run("num_records 8192 B, lane 3 voffset 8192-16+... + soffset 4096 in range of memory", 8192, 4096, 3, 8192 - 16);
  run("num_records 8192 B, lane 3 voffset 4096, soffset 4096 (sum = num_records)", 8192, 4096, 3, 4096);
  return 0;
}

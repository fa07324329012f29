// 8x8 block DCT pass (block_transform_k<8> of pointwise.hip: fp32 NHWC in -> fp32 NHWC out + bf16 copy) -- how much of its
// time is the width of its memory instructions?  V0 = the shipped mapping (one (block, channel) per thread, 4-byte loads and
// stores, 2-byte bf16 stores), V1 = two channels per thread (8-byte loads/stores, 4-byte bf16 stores), V2 = patch staged
// through LDS (16-byte loads/stores, 8-byte bf16 stores; the transform reads and writes its 64 values in LDS).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/dct_store.hip -o /tmp/dct_store ; run: /tmp/dct_store [B H W C]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
constexpr int BS = 8;

__device__ __forceinline__ void dct8(const float (&X)[BS][BS], const float* T, float (&Y)[BS][BS]) {
  float tmp[BS][BS];
#pragma unroll
  for (int u = 0; u < BS; ++u) {
    float tr[BS];
#pragma unroll
    for (int k = 0; k < BS; ++k) tr[k] = T[u * BS + k];
#pragma unroll
    for (int v = 0; v < BS; ++v) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < BS; ++k) a = fmaf(tr[k], X[k][v], a);
      tmp[u][v] = a;
    }
  }
#pragma unroll
  for (int v = 0; v < BS; ++v) {
    float tr[BS];
#pragma unroll
    for (int k = 0; k < BS; ++k) tr[k] = T[v * BS + k];
#pragma unroll
    for (int u = 0; u < BS; ++u) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < BS; ++k) a = fmaf(tmp[u][k], tr[k], a);
      Y[u][v] = a;
    }
  }
}

__global__ __launch_bounds__(128) void v0(const float* __restrict__ in, float* __restrict__ out, __bf16* __restrict__ o16, int B, int H, int W,
                                          int C, const float* __restrict__ mats) {
  __shared__ float Ts[64];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) Ts[i] = mats[i];
  __syncthreads();
  const int nby = H / BS, nbx = W / BS;
  const int64_t total = (int64_t)B * nby * nbx * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = i % C; int64_t p = i / C; int bx = p % nbx; p /= nbx; int by = p % nby; int b = p / nby;
    float X[BS][BS], Y[BS][BS];
#pragma unroll
    for (int u = 0; u < BS; ++u)
#pragma unroll
      for (int v = 0; v < BS; ++v) X[u][v] = in[(((size_t)b * H + by * BS + u) * W + bx * BS + v) * C + c];
    dct8(X, Ts, Y);
#pragma unroll
    for (int v = 0; v < BS; ++v)
#pragma unroll
      for (int u = 0; u < BS; ++u) {
        const size_t o = (((size_t)b * H + by * BS + u) * W + bx * BS + v) * C + c;
        out[o] = Y[u][v];
        o16[o] = (__bf16)Y[u][v];
      }
  }
}

__global__ __launch_bounds__(128) void v1(const float* __restrict__ in, float* __restrict__ out, __bf16* __restrict__ o16, int B, int H, int W,
                                          int C, const float* __restrict__ mats) {
  __shared__ float Ts[64];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) Ts[i] = mats[i];
  __syncthreads();
  const int nby = H / BS, nbx = W / BS, C2 = C / 2;
  const int64_t total = (int64_t)B * nby * nbx * C2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (i % C2) * 2; int64_t p = i / C2; int bx = p % nbx; p /= nbx; int by = p % nby; int b = p / nby;
    float X0[BS][BS], X1[BS][BS];
#pragma unroll
    for (int u = 0; u < BS; ++u)
#pragma unroll
      for (int v = 0; v < BS; ++v) {
        const float2 t = *reinterpret_cast<const float2*>(in + (((size_t)b * H + by * BS + u) * W + bx * BS + v) * C + c);
        X0[u][v] = t.x; X1[u][v] = t.y;
      }
    float Y0[BS][BS], Y1[BS][BS];
    dct8(X0, Ts, Y0);
    dct8(X1, Ts, Y1);
#pragma unroll
    for (int v = 0; v < BS; ++v)
#pragma unroll
      for (int u = 0; u < BS; ++u) {
        const size_t o = (((size_t)b * H + by * BS + u) * W + bx * BS + v) * C + c;
        *reinterpret_cast<float2*>(out + o) = make_float2(Y0[u][v], Y1[u][v]);
        bf16x2 h; h[0] = (__bf16)Y0[u][v]; h[1] = (__bf16)Y1[u][v];
        *reinterpret_cast<bf16x2*>(o16 + o) = h;
      }
  }
}

// V2: a workgroup of 128 threads owns NB = 128 / C spatial blocks x C channels (C = 64: two 8x8 blocks, 32 KiB of LDS).
// Patch layout in LDS: [block][pixel 0..63][C + 4 pad] floats (pad: the per-thread column walk stays conflict-free and the
// float4 row access keeps 16-byte alignment).
template <int C>
__global__ __launch_bounds__(128) void v2(const float* __restrict__ in, float* __restrict__ out, __bf16* __restrict__ o16, int B, int H, int W,
                                          const float* __restrict__ mats) {
  constexpr int NB = 128 / C, CP = C + 4, C4 = C / 4;
  __shared__ float Ts[64];
  __shared__ __attribute__((aligned(16))) float patch[NB * 64 * CP];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) Ts[i] = mats[i];
  const int nby = H / BS, nbx = W / BS;
  const int64_t nblk = (int64_t)B * nby * nbx;
  for (int64_t g = (int64_t)blockIdx.x * NB; g < nblk; g += (int64_t)gridDim.x * NB) {
    __syncthreads();
    // load: NB*64 pixels x C4 float4s
    for (int s = threadIdx.x; s < NB * 64 * C4; s += 128) {
      const int c4 = s % C4, px = (s / C4) % 64, nb = s / (C4 * 64);
      const int64_t blk = g + nb;
      if (blk >= nblk) continue;
      int64_t p = blk; const int bx = p % nbx; p /= nbx; const int by = p % nby; const int b = p / nby;
      const float4 t = *reinterpret_cast<const float4*>(in + (((size_t)b * H + by * BS + (px >> 3)) * W + bx * BS + (px & 7)) * C + 4 * c4);
      *reinterpret_cast<float4*>(patch + (nb * 64 + px) * CP + 4 * c4) = t;
    }
    __syncthreads();
    {
      const int c = threadIdx.x % C, nb = threadIdx.x / C;
      float* pp = patch + nb * 64 * CP + c;
      float X[BS][BS], Y[BS][BS];
#pragma unroll
      for (int u = 0; u < BS; ++u)
#pragma unroll
        for (int v = 0; v < BS; ++v) X[u][v] = pp[(u * 8 + v) * CP];
      dct8(X, Ts, Y);
#pragma unroll
      for (int u = 0; u < BS; ++u)
#pragma unroll
        for (int v = 0; v < BS; ++v) pp[(u * 8 + v) * CP] = Y[u][v];
    }
    __syncthreads();
    for (int s = threadIdx.x; s < NB * 64 * C4; s += 128) {
      const int c4 = s % C4, px = (s / C4) % 64, nb = s / (C4 * 64);
      const int64_t blk = g + nb;
      if (blk >= nblk) continue;
      int64_t p = blk; const int bx = p % nbx; p /= nbx; const int by = p % nby; const int b = p / nby;
      const float4 t = *reinterpret_cast<const float4*>(patch + (nb * 64 + px) * CP + 4 * c4);
      const size_t o = (((size_t)b * H + by * BS + (px >> 3)) * W + bx * BS + (px & 7)) * C + 4 * c4;
      *reinterpret_cast<float4*>(out + o) = t;
      bf16x4 h; h[0] = (__bf16)t.x; h[1] = (__bf16)t.y; h[2] = (__bf16)t.z; h[3] = (__bf16)t.w;
      *reinterpret_cast<bf16x4*>(o16 + o) = h;
    }
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 256, W = argc > 3 ? atoi(argv[3]) : 256, C = argc > 4 ? atoi(argv[4]) : 64;
  const size_t n = (size_t)B * H * W * C;
  float *in, *out, *mats; __bf16* o16;
  hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&o16, n * 2); hipMalloc(&mats, 256);
  std::vector<float> h(n), T(64);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) T[i * 8 + j] = i == 0 ? sqrtf(1.f / 8) : sqrtf(2.f / 8) * cosf(M_PI * (2 * j + 1) * i / 16);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(mats, T.data(), 256, hipMemcpyHostToDevice);
  std::vector<float> ref(n), got(n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto bench = [&](const char* name, auto launch) {
    hipMemset(out, 0, n * 4);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost);
    double md = 0;
    if (ref[0] == 0 && ref[1] == 0) ref = got; else for (size_t i = 0; i < n; i += 97) md = fmax(md, fabs(got[i] - ref[i]));
    printf("%-34s %8.1f us  %7.1f GB/s  (max diff vs V0 %.1e)\n", name, ms * 1e3, n * 10.0 / ms / 1e6, md);
  };
  printf("B=%d H=%d W=%d C=%d: %.0f MB in, %.0f MB out\n", B, H, W, C, n * 4 / 1e6, n * 6 / 1e6);
  const int64_t tot = (int64_t)B * (H / 8) * (W / 8) * C;
  auto grid = [](int64_t t, int bs) { int64_t g = (t + bs - 1) / bs; return (unsigned)(g > 65536 ? 65536 : g); };
  bench("V0 1 channel/thread (shipped)", [&] { hipLaunchKernelGGL(v0, dim3(grid(tot, 128)), dim3(128), 0, 0, in, out, o16, B, H, W, C, mats); });
  bench("V1 2 channels/thread", [&] { hipLaunchKernelGGL(v1, dim3(grid(tot / 2, 128)), dim3(128), 0, 0, in, out, o16, B, H, W, C, mats); });
  if (C == 64) bench("V2 LDS-staged patch, 16-byte I/O", [&] { hipLaunchKernelGGL(v2<64>, dim3(grid(tot / 128, 1)), dim3(128), 0, 0, in, out, o16, B, H, W, mats); });
  if (C == 128) bench("V2 LDS-staged patch, 16-byte I/O", [&] { hipLaunchKernelGGL(v2<128>, dim3(grid(tot / 128, 1)), dim3(128), 0, 0, in, out, o16, B, H, W, mats); });
  return 0;
}

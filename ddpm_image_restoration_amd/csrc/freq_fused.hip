// Frequency block of the DCT families in ONE pass (webp_inference.py:231-270, svd.ipynb c0:198-237), bf16 mode, C = 64:
//
//   xd  = D x D^T per bs x bs block and channel          (DCTLayer, webp_inference.py:154-203)
//   hid = lrelu(W1 bf16(xd) + b1)                         (first layers of the low and the high gate, stacked)
//   g   = sigmoid(W2 bf16(hid) + b2), the low or the high gate by the pixel's position in its block, high gate x boost[b]
//   S   = x + g * xd  -> bf16                             (operand of conv_out)
//
// The three-kernel path (block_transform_fast_k, two gemm16 launches) moves, per pixel and channel, 4 + 4 + 2 bytes for the
// DCT, 2 + 2 for the first layer and 2 + 4 + 4 + 2 for the second = 26 bytes; this kernel reads x once and writes S once
// (6 bytes): at 256 x 256, B = 16 that is 1.74 GB -> 0.40 GB per block and 490 us -> 172 us.
// Arithmetic is the same instruction sequence as the kernels it replaces (DCT FMAs in the same order, MFMA k-steps in the
// same order, the same roundings to bf16), so S comes out bit-identical (tests/test_gpu_ops.py).
//
// One workgroup = 128 pixels (two 8x8 or eight 4x4 blocks) x 64 channels, 4 waves; wave w owns pixel rows 32w .. 32w+31 of
// both GEMMs (rows are wave-private, so the hidden layer overwrites the wave's own operand rows in place).
// LDS (52 KiB, three workgroups per CU): fp32 patch [128][64 + 4] (x, then xd in place) | bf16 operand rows [128][128 B]
// with the XOR chunk swizzle of gemm16.hip; after the second GEMM the operand rows become the waves' staging tiles (two
// halves of 16 rows) for row-contiguous stores.  The weight fragments (24 KiB for all workgroups) come straight from
// global memory (WLDS = false): with W1 and the compact W2 staged in LDS the footprint is 68 KiB and only two workgroups
// fit, 192 us instead of 172 us for the level-0 tensor (B = 16, 256 x 256); the three kernels it replaces took 490 us.
#include "kernels.h"

namespace ddpmir {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int FC = 64, FPT = 128, FXS = FC + 4;
constexpr int F_X32_BYTES = FPT * FXS * 4;             // 34 816
constexpr int F_A16 = 0, F_W1 = 16384, F_W2 = 24576, F_R_BYTES = 32768;

template <int BS, bool WLDS>
__global__ __launch_bounds__(256, WLDS ? 2 : 3) void freq_gate_fused_k(const float* __restrict__ x, const float* __restrict__ dct,
                                                            const __bf16* __restrict__ w1, const float* __restrict__ b1,
                                                            const __bf16* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ boost, __bf16* __restrict__ S16, int H, int W,
                                                            int low) {
  constexpr int C = FC, BB = BS * BS, NBLK = FPT / BB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* X32 = reinterpret_cast<float*>(smem);
  unsigned char* R = smem + F_X32_BYTES;
  float* Ts = reinterpret_cast<float*>(R + (WLDS ? F_R_BYTES : F_W1));
  float* B1s = Ts + BB;
  float* B2s = B1s + C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbx = W / BS, nby = H / BS;
  // pixel row (0..127) of tile t -> NHWC pixel index (tiles do not straddle images: host check)
  auto pix = [&](int64_t t, int row) -> int64_t {
    const int j = row / BB, rem = row - j * BB, u = rem / BS, v = rem - u * BS;
    const int64_t gb = t * NBLK + j;
    const int bx = (int)(gb % nbx), by = (int)((gb / nbx) % nby), b = (int)(gb / ((int64_t)nbx * nby));
    return ((int64_t)b * H + by * BS + u) * W + bx * BS + v;
  };
  auto is_low = [&](int row) -> bool {
    const int rem = row % BB, u = rem / BS, v = rem - u * BS;
    return u < low && v < low;
  };

  // ---- phase A: x patch -> LDS (and kept in registers for the final S = x + g * xd), weights -> LDS
  const int64_t tile = blockIdx.x;
  const int c4 = lane & 15, rsub = lane >> 4;
  float4 xreg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 32 * wave + 4 * i + rsub;
    xreg[i] = *reinterpret_cast<const float4*>(x + pix(tile, row) * C + 4 * c4);
    *reinterpret_cast<float4*>(X32 + row * FXS + 4 * c4) = xreg[i];
  }
  if constexpr (WLDS)
#pragma unroll
  for (int q = tid; q < 512; q += 256) {
    { const int n = q >> 3, p = q & 7;                                   // W1 [64][64]: 8 chunks of 8 k per row
      *reinterpret_cast<bf16x8*>(R + F_W1 + n * 128 + ((p ^ ((n >> 1) & 7)) << 4)) = *reinterpret_cast<const bf16x8*>(w1 + n * C + 8 * p); }
    { const int n = q >> 2, p = q & 3;                                   // W2 [128][64] -> the 32 k its gate reads
      *reinterpret_cast<bf16x8*>(R + F_W2 + n * 64 + ((p ^ ((n >> 2) & 3)) << 4)) =
          *reinterpret_cast<const bf16x8*>(w2 + n * C + (n >= C ? C / 2 : 0) + 8 * p); }
  }
  if (tid < BB) Ts[tid] = dct[tid];
  if (tid < C) B1s[tid] = b1[tid];
  if (tid < 2 * C) B2s[tid] = b2[tid];
  __syncthreads();

  // ---- phase B: block DCT in place (fp32) + bf16 operand rows.  Same FMA order as block_transform_fast_k.
  auto put16 = [&](int row, int c, float v) {
    *reinterpret_cast<__bf16*>(R + F_A16 + row * 128 + (((c >> 3) ^ ((row >> 1) & 7)) << 4) + (c & 7) * 2) = (__bf16)v;
  };
  if constexpr (BS == 8) {
    // 128 (block, channel) pairs, two threads each: pass 1 = T X on four columns, pass 2 = (T X) T^T on four rows
    const int c = tid & 63, j = (tid >> 6) & 1, half = tid >> 7;
    float* base = X32 + (j * 64) * FXS + c;
    {
      float xv[8][4], tm[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[u][v] = base[(u * 8 + 4 * half + v) * FXS];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) a = fmaf(Ts[u * 8 + k], xv[k][v], a);
          tm[u][v] = a;
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) base[(u * 8 + 4 * half + v) * FXS] = tm[u][v];
    }
    __syncthreads();
    {
      float tm[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) tm[u][k] = base[((4 * half + u) * 8 + k) * FXS];
#pragma unroll
      for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) a = fmaf(tm[u][k], Ts[v * 8 + k], a);
          const int row = j * 64 + (4 * half + u) * 8 + v;
          X32[row * FXS + c] = a;
          put16(row, c, a);
        }
    }
  } else {
    // 512 (block, channel) pairs of 16 values: two whole 4x4 transforms per thread
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int pair = tid + 256 * pp, c = pair & 63, j = pair >> 6;
      float* base = X32 + (j * 16) * FXS + c;
      float xv[4][4], tm[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[u][v] = base[(u * 4 + v) * FXS];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fmaf(Ts[u * 4 + k], xv[k][v], a);
          tm[u][v] = a;
        }
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fmaf(tm[u][k], Ts[v * 4 + k], a);
          const int row = j * 16 + u * 4 + v;
          X32[row * FXS + c] = a;
          put16(row, c, a);
        }
    }
  }
  __syncthreads();

  // ---- phase C: hid = lrelu(W1 xd16 + b1), written over the wave's own operand rows
  const int lr = lane & 31, lh = lane >> 5;
  const int arow = 32 * wave + lr, a_key = (arow >> 1) & 7, w_key = (lr >> 1) & 7;
  f32x16 acc1[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[nt][r] = 0.f;
  {
    bf16x8 af[4], bfr[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      af[ks] = *reinterpret_cast<const bf16x8*>(R + F_A16 + arow * 128 + (((2 * ks + lh) ^ a_key) << 4));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bfr[ks][nt] = WLDS ? *reinterpret_cast<const bf16x8*>(R + F_W1 + (nt * 32 + lr) * 128 + (((2 * ks + lh) ^ w_key) << 4))
                           : *reinterpret_cast<const bf16x8*>(w1 + (nt * 32 + lr) * C + (2 * ks + lh) * 8);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfr[ks][nt], acc1[nt], 0, 0, 0);
  }
  __syncthreads();       // every lane of the wave has its operand fragments before the rows are overwritten
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = nt * 32 + lr;
    const float bc = B1s[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
      put16(row, col, act_apply(acc1[nt][r] + bc, ACT_LRELU));
    }
  }
  __syncthreads();

  // ---- phase D: gate pre-activations: low gate = W2[0:64] hid[0:32], high gate = W2[64:128] hid[32:64]
  f32x16 acc2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
  {
    bf16x8 hf[4], bw[2][4];
    const int k2 = (lr >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) hf[ks] = *reinterpret_cast<const bf16x8*>(R + F_A16 + arow * 128 + (((2 * ks + lh) ^ a_key) << 4));
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        bw[kk][nt] = WLDS ? *reinterpret_cast<const bf16x8*>(R + F_W2 + (nt * 32 + lr) * 64 + (((2 * kk + lh) ^ k2) << 4))
                          : *reinterpret_cast<const bf16x8*>(w2 + (nt * 32 + lr) * C + (nt >= 2 ? C / 2 : 0) + (2 * kk + lh) * 8);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[(nt >= 2 ? 2 : 0) + kk], bw[kk][nt], acc2[nt], 0, 0, 0);
  }
  __syncthreads();       // the operand rows are dead from here: that region becomes the waves' staging tiles

  // ---- phase E: per pixel keep its gate, S = x + sigmoid(.) * xd (* boost for the high gate), row-contiguous bf16 stores
  // (two halves of 16 rows per wave: 4 waves x 16 x 64 floats = the 16 KiB of the operand rows; the weights stay in LDS)
  float* st = reinterpret_cast<float*>(R + F_A16) + wave * (16 * 64);
  const float bs = boost[(int)(tile * NBLK / ((int64_t)nbx * nby))];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      const int r = 8 * hh + r8;
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;              // 16 hh .. 16 hh + 15
      const bool lowp = is_low(32 * wave + row);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int col = q * 32 + lr;
        st[(row - 16 * hh) * 64 + col] = lowp ? acc2[q][r] + B2s[col] : acc2[2 + q][r] + B2s[C + col];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int i = 4 * hh + i4, row = 4 * i + rsub, grow = 32 * wave + row;
      float4 v = *reinterpret_cast<const float4*>(st + (row - 16 * hh) * 64 + 4 * c4);
      const float4 xd = *reinterpret_cast<const float4*>(X32 + grow * FXS + 4 * c4);
      v.x = act_apply(v.x, ACT_SIGMOID); v.y = act_apply(v.y, ACT_SIGMOID); v.z = act_apply(v.z, ACT_SIGMOID); v.w = act_apply(v.w, ACT_SIGMOID);
      v.x *= xd.x; v.y *= xd.y; v.z *= xd.z; v.w *= xd.w;
      if (!is_low(grow)) { v.x *= bs; v.y *= bs; v.z *= bs; v.w *= bs; }
      v.x += xreg[i].x; v.y += xreg[i].y; v.z += xreg[i].z; v.w += xreg[i].w;
      bf16x4 o;
      o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(S16 + pix(tile, grow) * C + 4 * c4) = o;
    }
    __syncthreads();     // the staging rows are free for the second half
  }
}

template <int BS>
void launch_bs(const float* x, const float* dct, const void* w1, const float* b1, const void* w2, const float* b2, const float* boost,
               void* S16, int B, int H, int W, int low, hipStream_t s) {
  constexpr bool WLDS = false;      // weight fragments straight from global memory (L1/L2): 52 KiB of LDS, three workgroups per CU
  constexpr int LDS = F_X32_BYTES + (WLDS ? F_R_BYTES : F_W1) + (BS * BS + FC + 2 * FC) * 4;
  static PerDeviceOnce attr;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&freq_gate_fused_k<BS, WLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  });
  const int64_t tiles = (int64_t)B * (H / BS) * (W / BS) / (FPT / (BS * BS));
  // one tile per workgroup: a persistent walk over the tiles with the next tile's x prefetched into registers was measured
  // SLOWER (278 vs 192 us for the level-0 tensor): the hardware's own back-filling of finished workgroups overlaps the load
  // phase of one tile with the store phase of another better than a serial tile loop does
  hipLaunchKernelGGL((freq_gate_fused_k<BS, WLDS>), dim3((unsigned)tiles), dim3(256), LDS, s, x, dct, reinterpret_cast<const __bf16*>(w1), b1,
                     reinterpret_cast<const __bf16*>(w2), b2, boost, reinterpret_cast<__bf16*>(S16), H, W, low);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace

bool freq_gate_fused_supported(int B, int H, int W, int C, int bs) {
  if (C != FC || (bs != 4 && bs != 8) || B <= 0 || H <= 0 || W <= 0 || (H % bs) || (W % bs)) return false;
  const int per_image = (H / bs) * (W / bs), nblk = FPT / (bs * bs);
  return (per_image % nblk) == 0;                       // a 128-pixel tile never straddles two images
}

void launch_freq_gate_fused(const float* x, const float* dct, const void* w1, const float* b1, const void* w2, const float* b2,
                            const float* boost, void* S16, int B, int H, int W, int C, int bs, int low, hipStream_t s) {
  if (!freq_gate_fused_supported(B, H, W, C, bs)) throw std::runtime_error("fused frequency gates: unsupported shape");
  if (bs == 8) launch_bs<8>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, low, s);
  else launch_bs<4>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, low, s);
}

}  // namespace ddpmir

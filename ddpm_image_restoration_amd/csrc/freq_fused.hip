// Frequency block of the DCT families in ONE pass (webp_inference.py:231-270, svd.ipynb c0:198-237), bf16 mode, for the
// blocks with C = 64 or 128 channels (levels 0-1 of the 256 x 256 pyramid):
//
//   xd  = D x D^T per bs x bs block and channel          (DCTLayer, webp_inference.py:154-203)
//   hid = lrelu(W1 bf16(xd) + b1)                         (first layers of the low and the high gate, stacked)
//   g   = sigmoid(W2 bf16(hid) + b2), the low or the high gate by the pixel's position in its block, high gate x boost[b]
//   S   = x + g * xd  -> bf16                             (operand of conv_out)
//
// The three-kernel path (block_transform_fast_k, two gemm16 launches) moves, per pixel and channel, 4 + 4 + 2 bytes for the
// DCT, 2 + 2 for the first layer and 2 + 4 + 4 + 2 for the second = 26 bytes; this kernel reads x once and writes S once
// (6 bytes): at 256 x 256, B = 16, C = 64 that is 1.74 GB -> 0.40 GB per block and 490 us -> 172 us.
// Arithmetic is the same instruction sequence as the kernels it replaces (DCT FMAs in the same order, MFMA k-steps in the
// same order, the same roundings to bf16), so S comes out bit-identical (tests/test_gpu_ops.py).
//
// One workgroup = 8192 / C pixels (C = 64: two 8x8 or eight 4x4 blocks; C = 128: one or four) x C channels, 4 waves; a wave
// owns a 32-pixel x 64-channel quadrant of both GEMMs' outputs and of the final store.
// LDS (52 KiB, three workgroups per CU): fp32 patch [pixels][C + 4] (x, then xd in place) | bf16 operand rows [pixels][2C B]
// with an XOR chunk swizzle (gemm16.hip's rule); after the second GEMM the operand rows become the waves' staging tiles (two
// halves of 16 rows) for row-contiguous stores.  The weight fragments come straight from global memory (L1/L2): with W1
// and the compact W2 staged in LDS the footprint is 68 KiB, only two workgroups fit and the level-0 tensor (B = 16,
// 256 x 256 x 64) takes 192 us instead of 172 us; the three kernels this replaces took 490 us.  A persistent walk over the
// tiles with the next tile's x prefetched into registers was slower still (278 us): the hardware's back-filling of finished
// workgroups overlaps the load phase of one tile with the store phase of another better than a serial tile loop.
#include "kernels.h"

namespace ddpmir {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int F_ELEMS = 8192;                          // pixels x channels of one tile

template <int C>
struct FGeom {
  static constexpr int PT = F_ELEMS / C;               // pixels per tile: 128 or 64
  static constexpr int XS = C + 4;                     // fp32 patch row stride (floats)
  static constexpr int X32_BYTES = PT * XS * 4;
  static constexpr int ROWB = 2 * C;                   // bf16 operand row bytes: 128 or 256
  static constexpr int CPR = C / 8;                    // 16-byte chunks per operand row
  static constexpr int RSH = C == 64 ? 1 : 0;          // operand rows per 256-byte bank row = 1 << RSH
  static constexpr int A16_BYTES = PT * ROWB;          // 16 KiB
  static constexpr int KS = C / 16;                    // k-steps of the first layer
  static constexpr int KK = C / 32;                    // k-steps of one gate's second layer (it reads half the hidden units)
};

// OP = true: x is not read as it stands but formed first as  x = W_o att + b_o + x_in  (the attention's output projection with
// its residual, webp_inference.py:317-321: MultiheadAttention.out_proj and h + attn) -- one more GEMM on the tile instead of
// a separate pass that writes x (fp32) only for this kernel to read it back.
template <int BS, int C, bool OP>
__global__ __launch_bounds__(256, 3) void freq_gate_fused_k(const float* __restrict__ x, const float* __restrict__ dct,
                                                            const __bf16* __restrict__ w1, const float* __restrict__ b1,
                                                            const __bf16* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ boost, __bf16* __restrict__ S16, int H, int W,
                                                            int low, const __bf16* __restrict__ att16, const __bf16* __restrict__ wo,
                                                            const float* __restrict__ bo, unsigned char* __restrict__ S8,
                                                            const __bf16* __restrict__ xin16) {
  using G = FGeom<C>;
  constexpr int BB = BS * BS, NBLK = G::PT / BB, XS = G::XS, ROWB = G::ROWB, CPR = G::CPR, RSH = G::RSH, KS = G::KS, KK = G::KK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* X32 = reinterpret_cast<float*>(smem);
  unsigned char* A16 = smem + G::X32_BYTES;
  float* Ts = reinterpret_cast<float*>(A16 + G::A16_BYTES);
  float* B1s = Ts + BB;
  float* B2s = B1s + C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbx = W / BS, nby = H / BS;
  const int64_t tile = blockIdx.x;
  // pixel row of the tile -> NHWC pixel index (tiles do not straddle images: host check)
  auto pix = [&](int row) -> int64_t {
    const int j = row / BB, rem = row - j * BB, u = rem / BS, v = rem - u * BS;
    const int64_t gb = tile * NBLK + j;
    const int bx = (int)(gb % nbx), by = (int)((gb / nbx) % nby), b = (int)(gb / ((int64_t)nbx * nby));
    return ((int64_t)b * H + by * BS + u) * W + bx * BS + v;
  };
  auto is_low = [&](int row) -> bool {
    const int rem = row % BB, u = rem / BS, v = rem - u * BS;
    return u < low && v < low;
  };
  // the wave's quadrant: 32 pixel rows x 64 channels
  const int row0 = C == 64 ? 32 * wave : 32 * (wave & 1);
  const int col0 = C == 64 ? 0 : 64 * (wave >> 1);

  // ---- phase A: x patch -> LDS (and kept in registers for the final S = x + g * xd)
  const int c4 = lane & 15, rsub = lane >> 4;
  float4 xreg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = row0 + 4 * i + rsub;
    if (xin16) {   // x arrives as bf16 (the producing conv wrote no fp32 copy: half the bytes of this kernel's read)
      const bf16x4 t = *reinterpret_cast<const bf16x4*>(xin16 + pix(row) * C + col0 + 4 * c4);
      xreg[i] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    } else {
      xreg[i] = *reinterpret_cast<const float4*>(x + pix(row) * C + col0 + 4 * c4);
    }
    *reinterpret_cast<float4*>(X32 + row * XS + col0 + 4 * c4) = xreg[i];
  }
  if (tid < BB) Ts[tid] = dct[tid];
  if (tid < C) B1s[tid] = b1[tid];
  if (tid < 2 * C) B2s[tid] = b2[tid];
  const int lr = lane & 31, lh = lane >> 5;
  const int arow = row0 + lr, a_key = (arow >> RSH) & (CPR - 1);
  const unsigned char* arow_p = A16 + arow * ROWB;
  if constexpr (OP) {
    // attention output rows -> operand rows (16-byte chunks, swizzled), then x = W_o att + b_o + x_in in the fp32 patch
    if (tid < C) B2s[2 * C + tid] = bo[tid];
    constexpr int CH = G::PT * CPR / 256;              // chunks per thread: 4
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int q = i * 256 + tid, row = q / CPR, p = q % CPR;
      *reinterpret_cast<bf16x8*>(A16 + row * ROWB + ((p ^ ((row >> RSH) & (CPR - 1))) << 4)) =
          *reinterpret_cast<const bf16x8*>(att16 + pix(row) * C + 8 * p);
    }
    __syncthreads();
    f32x16 acc0[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[nt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow_p + (((2 * ks + lh) ^ a_key) << 4));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(wo + (size_t)(col0 + nt * 32 + lr) * C + (2 * ks + lh) * 8);
        acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc0[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = col0 + nt * 32 + lr;
      const float bc = B2s[2 * C + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* px = X32 + (row0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * XS + col;
        float v = acc0[nt][r];
        v += bc;
        v += *px;
        *px = v;
      }
    }
    __syncthreads();     // x is complete; every wave is done with the operand rows
#pragma unroll
    for (int i = 0; i < 8; ++i) xreg[i] = *reinterpret_cast<const float4*>(X32 + (row0 + 4 * i + rsub) * XS + col0 + 4 * c4);
  }
  __syncthreads();

  // ---- phase B: block DCT in place (fp32) + bf16 operand rows.  Same FMA order as block_transform_fast_k.
  auto put16 = [&](int row, int c, float v) {
    *reinterpret_cast<__bf16*>(A16 + row * ROWB + (((c >> 3) ^ ((row >> RSH) & (CPR - 1))) << 4) + (c & 7) * 2) = (__bf16)v;
  };
  if constexpr (BS == 8) {
    // 128 (block, channel) pairs, two threads each: pass 1 = T X on four columns, pass 2 = (T X) T^T on four rows
    const int pair = tid & 127, c = pair % C, j = pair / C, half = tid >> 7;
    float* base = X32 + (j * 64) * XS + c;
    {
      float xv[8][4], tm[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[u][v] = base[(u * 8 + 4 * half + v) * XS];
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
        for (int v = 0; v < 4; ++v) base[(u * 8 + 4 * half + v) * XS] = tm[u][v];
    }
    __syncthreads();
    {
      float tm[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) tm[u][k] = base[((4 * half + u) * 8 + k) * XS];
#pragma unroll
      for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) a = fmaf(tm[u][k], Ts[v * 8 + k], a);
          const int row = j * 64 + (4 * half + u) * 8 + v;
          X32[row * XS + c] = a;
          put16(row, c, a);
        }
    }
  } else {
    // 512 (block, channel) pairs of 16 values: two whole 4x4 transforms per thread
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int pair = tid + 256 * pp, c = pair % C, j = pair / C;
      float* base = X32 + (j * 16) * XS + c;
      float xv[4][4], tm[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[u][v] = base[(u * 4 + v) * XS];
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
          X32[row * XS + c] = a;
          put16(row, c, a);
        }
    }
  }
  __syncthreads();

  // ---- phase C: hid = lrelu(W1 xd16 + b1); the hidden layer then overwrites the operand rows
  f32x16 acc1[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[nt][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow_p + (((2 * ks + lh) ^ a_key) << 4));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(w1 + (size_t)(col0 + nt * 32 + lr) * C + (2 * ks + lh) * 8);
      acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc1[nt], 0, 0, 0);
    }
  }
  __syncthreads();       // every wave has read its operand rows before anybody overwrites them
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = col0 + nt * 32 + lr;
    const float bc = B1s[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) put16(row0 + (r & 3) + 8 * (r >> 2) + 4 * lh, col, act_apply(acc1[nt][r] + bc, ACT_LRELU));
  }
  __syncthreads();

  // ---- phase D: gate pre-activations of the wave's 64 channels: low gate = W2[c] hid[0:C/2], high gate = W2[C + c] hid[C/2:C]
  f32x16 acc2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g)                          // 0 = low gate, 1 = high gate
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int ks = g * KK + kk;                      // the hidden units this gate reads
      const bf16x8 hf = *reinterpret_cast<const bf16x8*>(arow_p + (((2 * ks + lh) ^ a_key) << 4));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bf16x8 bw = *reinterpret_cast<const bf16x8*>(w2 + (size_t)(g * C + col0 + q * 32 + lr) * C + (2 * ks + lh) * 8);
        acc2[2 * g + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, bw, acc2[2 * g + q], 0, 0, 0);
      }
    }
  __syncthreads();       // the operand rows are dead from here: that region becomes the waves' staging tiles

  // ---- phase E: per pixel keep its gate, S = x + sigmoid(.) * xd (* boost for the high gate), row-contiguous bf16 stores
  // (two halves of 16 rows per wave: 4 waves x 16 x 64 floats = the 16 KiB of the operand rows)
  float* st = reinterpret_cast<float*>(A16) + wave * (16 * 64);
  const float bs = boost[(int)(tile * NBLK / ((int64_t)nbx * nby))];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      const int r = 8 * hh + r8;
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;              // 16 hh .. 16 hh + 15
      const bool lowp = is_low(row0 + row);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int col = q * 32 + lr;
        st[(row - 16 * hh) * 64 + col] = lowp ? acc2[q][r] + B2s[col0 + col] : acc2[2 + q][r] + B2s[C + col0 + col];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int i = 4 * hh + i4, row = 4 * i + rsub, grow = row0 + row;
      float4 v = *reinterpret_cast<const float4*>(st + (row - 16 * hh) * 64 + 4 * c4);
      const float4 xd = *reinterpret_cast<const float4*>(X32 + grow * XS + col0 + 4 * c4);
      v.x = act_apply(v.x, ACT_SIGMOID); v.y = act_apply(v.y, ACT_SIGMOID); v.z = act_apply(v.z, ACT_SIGMOID); v.w = act_apply(v.w, ACT_SIGMOID);
      v.x *= xd.x; v.y *= xd.y; v.z *= xd.z; v.w *= xd.w;
      if (!is_low(grow)) { v.x *= bs; v.y *= bs; v.z *= bs; v.w *= bs; }
      v.x += xreg[i].x; v.y += xreg[i].y; v.z += xreg[i].z; v.w += xreg[i].w;
      if (S8) {   // operand of an e4m3 conv_out (DDPMIR_FP8)
        *reinterpret_cast<unsigned*>(S8 + pix(grow) * C + col0 + 4 * c4) = pack_e4m3x4(v);
      } else {
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(S16 + pix(grow) * C + col0 + 4 * c4) = o;
      }
    }
    __syncthreads();     // the staging rows are free for the second half
  }
}

template <int BS, int C, bool OP>
void launch_t(const float* x, const float* dct, const void* w1, const float* b1, const void* w2, const float* b2, const float* boost,
              void* S16, int B, int H, int W, int low, const void* att16, const void* wo, const float* bo, hipStream_t s, void* S8,
              const void* xin16) {
  using G = FGeom<C>;
  constexpr int LDS = G::X32_BYTES + G::A16_BYTES + (BS * BS + C + 2 * C + (OP ? C : 0)) * 4;
  static PerDeviceOnce attr;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&freq_gate_fused_k<BS, C, OP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  });
  const int64_t tiles = (int64_t)B * (H / BS) * (W / BS) / (G::PT / (BS * BS));
  hipLaunchKernelGGL((freq_gate_fused_k<BS, C, OP>), dim3((unsigned)tiles), dim3(256), LDS, s, x, dct, reinterpret_cast<const __bf16*>(w1), b1,
                     reinterpret_cast<const __bf16*>(w2), b2, boost, reinterpret_cast<__bf16*>(S16), H, W, low,
                     reinterpret_cast<const __bf16*>(att16), reinterpret_cast<const __bf16*>(wo), bo, reinterpret_cast<unsigned char*>(S8),
                     reinterpret_cast<const __bf16*>(xin16));
  DDPMIR_HIP_CHECK(hipGetLastError());
}

template <bool OP>
void launch_op(const float* x, const float* dct, const void* w1, const float* b1, const void* w2, const float* b2, const float* boost,
               void* S16, int B, int H, int W, int C, int bs, int low, const void* att16, const void* wo, const float* bo, hipStream_t s, void* S8,
               const void* xin16) {
  if (C == 64) {
    if (bs == 8) launch_t<8, 64, OP>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, low, att16, wo, bo, s, S8, xin16);
    else launch_t<4, 64, OP>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, low, att16, wo, bo, s, S8, xin16);
  } else {
    if (bs == 8) launch_t<8, 128, OP>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, low, att16, wo, bo, s, S8, xin16);
    else launch_t<4, 128, OP>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, low, att16, wo, bo, s, S8, xin16);
  }
}

}  // namespace

bool freq_gate_fused_supported(int B, int H, int W, int C, int bs) {
  if ((C != 64 && C != 128) || (bs != 4 && bs != 8) || B <= 0 || H <= 0 || W <= 0 || (H % bs) || (W % bs)) return false;
  const int per_image = (H / bs) * (W / bs), nblk = (F_ELEMS / C) / (bs * bs);
  return nblk >= 1 && (per_image % nblk) == 0;          // a tile never straddles two images
}

// att16 != nullptr: x is formed in the kernel as W_o att16 + b_o + x (the output projection of the block's attention)
void launch_freq_gate_fused(const float* x, const float* dct, const void* w1, const float* b1, const void* w2, const float* b2,
                            const float* boost, void* S16, int B, int H, int W, int C, int bs, int low, hipStream_t s,
                            const void* att16, const void* wo, const float* bo, void* S8, const void* xin16) {
  if (!S16 && !S8) throw std::runtime_error("fused frequency gates: no output");
  if (!x && !xin16) throw std::runtime_error("fused frequency gates: no input");
  if (!freq_gate_fused_supported(B, H, W, C, bs)) throw std::runtime_error("fused frequency gates: unsupported shape");
  if (att16) {
    if (!wo || !bo) throw std::runtime_error("fused frequency gates: output projection needs its weight and bias");
    launch_op<true>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, C, bs, low, att16, wo, bo, s, S8, xin16);
  } else {
    launch_op<false>(x, dct, w1, b1, w2, b2, boost, S16, B, H, W, C, bs, low, nullptr, nullptr, nullptr, s, S8, xin16);
  }
}

}  // namespace ddpmir

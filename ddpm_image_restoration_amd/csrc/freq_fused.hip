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
//
// What bounds it (tools/micro/freq_phase.hip: s_memtime stamps at the phase boundaries, 512 x 512, B = 8, bf16 in / out): 92 % of
// the 768 workgroup slots are occupied, the shader clock sits at 1.9-2.0 GHz, and a tile lives 15 us of which 5.5 us is the
// wait for its x (2.5 us even when x is L2-resident: kernel arguments, index arithmetic and the first instructions of three
// co-resident waves per SIMD), the rest VALU issue -- three waves per SIMD x ~1500 vector instructions x 4 cycles is 70 % of
// the tile's lifetime.  So the instruction count is the lever, and round 4 took the C = 64 kernel from 3744 to 2074
// instructions (466 -> 360 us in that microbenchmark, configs[4] bf16 10.96 -> 10.59 ms per timestep on one box):
//   * pixel addresses: block origins once per thread in 32-bit arithmetic instead of three 64-bit divisions per pixel row
//     (834 -> 275 instructions in phase A; the 4 x 4-block kernels went from ~9000 to ~1700 instructions);
//   * operand-row stores with compile-time swizzle keys and immediate offsets; a select instead of a divergent branch per
//     row in the gate staging; sigmoid through v_rcp_f32 (common.h sigmoid_fast);
//   * C = 64: the weight fragments of each gate layer are fetched one phase ahead (registers allow it), and the barriers
//     after that point order LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier), so neither those fetches nor the S stores
//     of the first half of phase E are waited for at a barrier.
// Tried and dropped: the DCT's FMA chains as v_pk_fma_f32 with op_sel broadcasts -- see fma2() below.
#include "kernels.h"

namespace ddpmir {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

namespace {

constexpr int F_ELEMS = 8192;                          // pixels x channels of one tile

// Two FMA chains side by side.  Written for v_pk_fma_f32 (the kernel is bound by VALU issue and the DCT is a quarter of its
// instructions), but with __builtin_elementwise_fma the kernel's output stopped being reproducible as soon as two workgroups
// shared a CU (tools/freq_fused_big.py: thousands of wrong values per launch at 256 x 256, none with one workgroup per CU, none
// with scalar FMAs; the ISA's wait counts look right) -- so these stay two v_fma_f32, and tests/test_gpu_ops.py checks the
// kernel at sizes with several workgroups per CU.
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; }

#ifdef DDPMIR_FREQ_PROBE
__device__ unsigned long long* g_probe;
// stamps stay in registers until the end of the tile: a store per stamp would put a store round trip into every phase
// (the workgroup barrier waits for outstanding stores)
#define PROBE(k) do { probe_t[k] = __builtin_amdgcn_s_memtime(); if ((k) == 8 && threadIdx.x == 0) for (int q_ = 0; q_ < 9; ++q_) g_probe[(size_t)blockIdx.x * 32 + q_] = probe_t[q_]; } while (0)
// per wave: entry and 'x loads returned' stamps (slots 16 + 2 wave, 17 + 2 wave)
#define PROBE_WAVE(slot, val) do { if ((threadIdx.x & 63) == 0) g_probe[(size_t)blockIdx.x * 32 + 16 + 2 * (threadIdx.x >> 6) + (slot)] = (val); } while (0)
#define PROBE_DECL unsigned long long probe_t[9]; const unsigned long long probe_r0 = __builtin_amdgcn_s_memrealtime();
#else
#define PROBE(k)
#define PROBE_DECL
#define PROBE_WAVE(slot, val)
#endif

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
  const unsigned nbx = W / BS, nby = H / BS;
  const unsigned tile = blockIdx.x;
  // Block j of the tile -> its image and the NHWC index of its top-left pixel.  32-bit arithmetic, once per block a thread
  // touches: as 64-bit divisions per pixel row this index arithmetic was a fifth of the kernel's instructions (the kernel is
  // bound by instruction issue, tools/micro/freq_phase.hip).  Tiles do not straddle images and B H W < 2^31: host checks.
  auto origin = [&](int j, int& b) -> int {
    const unsigned gb = tile * NBLK + j, t = gb / nbx, bx = gb - t * nbx;
    b = (int)(t / nby);
    const unsigned by = t - (unsigned)b * nby;
    return (int)(((unsigned)b * H + by * BS) * W + bx * BS);
  };
  auto is_low = [&](int row) -> bool {
    const int rem = row % BB, u = rem / BS, v = rem - u * BS;
    return u < low && v < low;
  };
  // Barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access of the wave: the weight
  // fragments fetched a phase ahead, the S stores of the first half of phase E -- round trips these barriers need not sit out.
  auto lds_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // the wave's quadrant: 32 pixel rows x 64 channels
  const int row0 = C == 64 ? 32 * wave : 32 * (wave & 1);
  const int col0 = C == 64 ? 0 : 64 * (wave >> 1);
  // a thread's rows of phases A and E are row0 + 4 i + rsub, i < 8: one block (8 x 8 blocks) or two (4 x 4) of the tile
  constexpr int JW = BB >= 32 ? 1 : 32 / BB;
  int org[JW], img_b = 0;
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) org[jj] = origin(row0 / BB + jj, img_b);
  auto pix = [&](int i, int rsub_) -> int {             // row0 + 4 i + rsub_, i a compile-time constant after unrolling
    const int jj = JW == 1 ? 0 : (4 * i) / BB;         // BB < 32: row0 is a multiple of 32, so of BB; rsub_ < 4 never carries
    const int rem = (row0 % BB) + 4 * i - jj * BB + rsub_;
    return org[jj] + (rem / BS) * W + rem % BS;
  };

  PROBE_DECL
  PROBE(0);
  // ---- phase A: x patch -> LDS (and kept in registers for the final S = x + g * xd)
  const int c4 = lane & 15, rsub = lane >> 4;
  float4 xreg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = row0 + 4 * i + rsub;
    if (xin16) {   // x arrives as bf16 (the producing conv wrote no fp32 copy: half the bytes of this kernel's read)
      const bf16x4 t = *reinterpret_cast<const bf16x4*>(xin16 + (int64_t)pix(i, rsub) * C + col0 + 4 * c4);
      xreg[i] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    } else {
      xreg[i] = *reinterpret_cast<const float4*>(x + (int64_t)pix(i, rsub) * C + col0 + 4 * c4);
    }
    *reinterpret_cast<float4*>(X32 + row * XS + col0 + 4 * c4) = xreg[i];
  }
#ifdef DDPMIR_FREQ_PROBE
  {
    const unsigned long long t_ld = __builtin_amdgcn_s_memtime();   // the LDS stores above waited for the loads
    PROBE_WAVE(0, probe_t[0]);
    PROBE_WAVE(1, t_ld);
  }
#endif
  if (tid < BB) Ts[tid] = dct[tid];
  if (tid < C) B1s[tid] = b1[tid];
  if (tid < 2 * C) B2s[tid] = b2[tid];
  const int lr = lane & 31, lh = lane >> 5;
  const int arow = row0 + lr, a_key = (arow >> RSH) & (CPR - 1);
  const unsigned char* arow_p = A16 + arow * ROWB;
  if constexpr (OP) {
    // attention output rows -> operand rows (16-byte chunks, swizzled), then x = W_o att + b_o + x_in in the fp32 patch
    if (tid < C) B2s[2 * C + tid] = bo[tid];
    int* orgs = reinterpret_cast<int*>(B2s + 3 * C);   // the operand rows are any rows of the tile: block origins through LDS
    if (tid < NBLK) { int b_; orgs[tid] = origin(tid, b_); }
    __syncthreads();
    auto pix_any = [&](int row) -> int64_t {
      const int rem = row % BB;
      return (int64_t)orgs[row / BB] + (rem / BS) * W + rem % BS;
    };
    constexpr int CH = G::PT * CPR / 256;              // chunks per thread: 4
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int q = i * 256 + tid, row = q / CPR, p = q % CPR;
      *reinterpret_cast<bf16x8*>(A16 + row * ROWB + ((p ^ ((row >> RSH) & (CPR - 1))) << 4)) =
          *reinterpret_cast<const bf16x8*>(att16 + pix_any(row) * C + 8 * p);
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
  PROBE(1);

  // ---- phase B: block DCT in place (fp32) + bf16 operand rows.  Same FMA order as block_transform_fast_k.
  // operand-row stores: row = rb + rc with rb a multiple of 32 (runtime) and rc a compile-time constant after unrolling, so the
  // row's swizzle key is the constant's -- one XOR per distinct key and immediate offsets instead of the address arithmetic per
  // element (cc = the channel's chunk << 4, cl = its byte in the chunk)
  auto put16 = [&](unsigned char* rowbase, int rc, int cc, float v) {
    *reinterpret_cast<__bf16*>(rowbase + rc * ROWB + (cc ^ (((rc >> RSH) & (CPR - 1)) << 4))) = (__bf16)v;
  };
  // C = 64: the weight fragments of both gate layers (32 registers each) are fetched a phase ahead of their GEMM, so the L2
  // round trip runs under the DCT / the hidden-row stores instead of in front of the MFMAs; C = 128 has no registers for that
  constexpr bool PRE = C == 64;
  bf16x8 w1f[PRE ? KS : 1][2], w2f[2][PRE ? KK : 1][2];
  auto fetch_w1 = [&]() {
    if constexpr (PRE) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w1f[ks][nt] = *reinterpret_cast<const bf16x8*>(w1 + (size_t)(col0 + nt * 32 + lr) * C + (2 * ks + lh) * 8);
    }
  };
  if constexpr (BS != 8) fetch_w1();
  if constexpr (BS == 8) {
    // 128 (block, channel) pairs, two threads each: pass 1 = T X on four columns, pass 2 = (T X) T^T on four rows
    const int pair = tid & 127, c = pair % C, j = pair / C, half = tid >> 7;
    float* base = X32 + (j * 64) * XS + c;
    unsigned char* ob = A16 + (j * 64 + 32 * half) * ROWB + (c & 7) * 2;
    float* xb = X32 + (j * 64 + 32 * half) * XS + c;
    const int cc = (c >> 3) << 4;
    {
      float xv[8][4], tm[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[u][v] = base[(u * 8 + 4 * half + v) * XS];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; v += 2) {
          f32x2 a = {0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 8; ++k) a = fma2(f32x2{Ts[u * 8 + k], Ts[u * 8 + k]}, f32x2{xv[k][v], xv[k][v + 1]}, a);
          tm[u][v] = a[0];
          tm[u][v + 1] = a[1];
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) base[(u * 8 + 4 * half + v) * XS] = tm[u][v];
    }
    __syncthreads();
    PROBE(2);
    fetch_w1();
    {
      float tm[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) tm[u][k] = base[((4 * half + u) * 8 + k) * XS];
#pragma unroll
      for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
          f32x2 a = {0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 8; ++k) a = fma2(f32x2{tm[u][k], tm[u + 1][k]}, f32x2{Ts[v * 8 + k], Ts[v * 8 + k]}, a);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int rc = (u + e) * 8 + v;              // row j * 64 + 32 * half + rc
            xb[rc * XS] = a[e];
            put16(ob, rc, cc, a[e]);
          }
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
        for (int v = 0; v < 4; v += 2) {
          f32x2 a = {0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fma2(f32x2{Ts[u * 4 + k], Ts[u * 4 + k]}, f32x2{xv[k][v], xv[k][v + 1]}, a);
          tm[u][v] = a[0];
          tm[u][v + 1] = a[1];
        }
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
          f32x2 a = {0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fma2(f32x2{tm[u][k], tm[u + 1][k]}, f32x2{Ts[v * 4 + k], Ts[v * 4 + k]}, a);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int row = j * 16 + (u + e) * 4 + v;
            X32[row * XS + c] = a[e];
            *reinterpret_cast<__bf16*>(A16 + row * ROWB + (((c >> 3) ^ ((row >> RSH) & (CPR - 1))) << 4) + (c & 7) * 2) = (__bf16)a[e];
          }
        }
    }
  }
  __syncthreads();
  PROBE(3);

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
      bf16x8 bfr;
      if constexpr (PRE) bfr = w1f[ks][nt];
      else bfr = *reinterpret_cast<const bf16x8*>(w1 + (size_t)(col0 + nt * 32 + lr) * C + (2 * ks + lh) * 8);
      acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc1[nt], 0, 0, 0);
    }
  }
  if constexpr (PRE) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          w2f[g][kk][q] = *reinterpret_cast<const bf16x8*>(w2 + (size_t)(g * C + col0 + q * 32 + lr) * C + (2 * (g * KK + kk) + lh) * 8);
  }
  lds_sync();            // every wave has read its operand rows before anybody overwrites them
  PROBE(4);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = col0 + nt * 32 + lr;
    const float bc = B1s[col];
    // row = row0 + 4 lh + rc: the key of 4 lh (bit 2 of the row) and of the constant rc (bits 0, 1, 3, 4) do not overlap
    unsigned char* hb = A16 + (row0 + 4 * lh) * ROWB + (col & 7) * 2;
    const int cc = ((col >> 3) ^ (((4 * lh) >> RSH) & (CPR - 1))) << 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) put16(hb, (r & 3) + 8 * (r >> 2), cc, act_apply(acc1[nt][r] + bc, ACT_LRELU));
  }
  lds_sync();
  PROBE(5);

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
        bf16x8 bw;
        if constexpr (PRE) bw = w2f[g][kk][q];
        else bw = *reinterpret_cast<const bf16x8*>(w2 + (size_t)(g * C + col0 + q * 32 + lr) * C + (2 * ks + lh) * 8);
        acc2[2 * g + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, bw, acc2[2 * g + q], 0, 0, 0);
      }
    }
  lds_sync();            // the operand rows are dead from here: that region becomes the waves' staging tiles
  PROBE(6);

  // ---- phase E: per pixel keep its gate, S = x + sigmoid(.) * xd (* boost for the high gate), row-contiguous bf16 stores
  // (two halves of 16 rows per wave: 4 waves x 16 x 64 floats = the 16 KiB of the operand rows)
  float* st = reinterpret_cast<float*>(A16) + wave * (16 * 64);
  const float bs = boost[img_b];
  const float b2lo[2] = {B2s[col0 + lr], B2s[col0 + 32 + lr]}, b2hi[2] = {B2s[C + col0 + lr], B2s[C + col0 + 32 + lr]};
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      const int r = 8 * hh + r8;
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;              // 16 hh .. 16 hh + 15
      const bool lowp = is_low(row0 + row);
#pragma unroll
      for (int q = 0; q < 2; ++q)                                    // both sums, then a select: no divergent branch per row
        st[(row - 16 * hh) * 64 + q * 32 + lr] = lowp ? acc2[q][r] + b2lo[q] : acc2[2 + q][r] + b2hi[q];
    }
    lds_sync();
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int i = 4 * hh + i4, row = 4 * i + rsub, grow = row0 + row;
      float4 v = *reinterpret_cast<const float4*>(st + (row - 16 * hh) * 64 + 4 * c4);
      const float4 xd = *reinterpret_cast<const float4*>(X32 + grow * XS + col0 + 4 * c4);
      v.x = sigmoid_fast(v.x); v.y = sigmoid_fast(v.y); v.z = sigmoid_fast(v.z); v.w = sigmoid_fast(v.w);   // = act16(., ACT_SIGMOID) of the GEMM epilogue
      v.x *= xd.x; v.y *= xd.y; v.z *= xd.z; v.w *= xd.w;
      if (!is_low(grow)) { v.x *= bs; v.y *= bs; v.z *= bs; v.w *= bs; }
      v.x += xreg[i].x; v.y += xreg[i].y; v.z += xreg[i].z; v.w += xreg[i].w;
      if (S8) {   // operand of an e4m3 conv_out (DDPMIR_FP8)
        *reinterpret_cast<unsigned*>(S8 + (int64_t)pix(i, rsub) * C + col0 + 4 * c4) = pack_e4m3x4(v);
      } else {
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
#ifdef DDPMIR_FREQ_PROBE_NOSTORE
        if (v.x == 12345.678f)       // experiment of tools/micro/freq_phase.hip: the kernel without its stores
#endif
        *reinterpret_cast<bf16x4*>(S16 + (int64_t)pix(i, rsub) * C + col0 + 4 * c4) = o;
      }
    }
    if (hh == 0) lds_sync();     // the staging rows are free for the second half
    if (hh == 0) PROBE(7);
  }
  PROBE(8);
#ifdef DDPMIR_FREQ_PROBE
  if (threadIdx.x == 0) { g_probe[(size_t)blockIdx.x * 32 + 24] = probe_r0; g_probe[(size_t)blockIdx.x * 32 + 25] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

template <int BS, int C, bool OP>
void launch_t(const float* x, const float* dct, const void* w1, const float* b1, const void* w2, const float* b2, const float* boost,
              void* S16, int B, int H, int W, int low, const void* att16, const void* wo, const float* bo, hipStream_t s, void* S8,
              const void* xin16) {
  using G = FGeom<C>;
  constexpr int LDS = G::X32_BYTES + G::A16_BYTES + (BS * BS + C + 2 * C + (OP ? C + 8 : 0)) * 4;
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

// Implicit-GEMM convolution / linear kernel, exact fp32 on the gfx950 matrix cores.
//
//   out[m][n] = epi( sum_{tap} sum_{c} pro(A[pixel(m) + tap][c]) * Wt[tap][n][c] )
//
// M = B*H*W output pixels (NHWC rows), N = Cout, K = taps*Cin.  One workgroup (4 waves) owns a
// 128 x (32*NT) output tile; each wave owns 32 rows x 32*NT columns as NT 32x32 accumulators of
// v_mfma_f32_32x32x2_f32 (bit-exact fp32 fma chain, MI355X_MICROARCH "FP32-input MFMA").  K is walked in
// 32-channel chunks per tap; the next chunk's global loads are issued into registers before the current
// chunk's MFMAs (issue-early / write-late staging) and the GroupNorm affine + activation of the producer
// layer is applied on the way into LDS, so normalised tensors are never materialised in HBM.
//
// LDS image: row-major [row][32 k + 4 pad] floats; a lane reads 16 B = 4 consecutive k of its row
// (ds_read_b128; the 36-float row stride is conflict-free for the 16-lane b128 groups).  Within a chunk
// the k order seen by the MFMA is permuted (lane half h takes k = 8g+4h+j) -- identical for A and B.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace ddpmir {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LROW = BK + 4;

template <int NT>
__global__ __launch_bounds__(256) void igemm_kernel(const IGemm p) {
  __shared__ __attribute__((aligned(16))) float As[BM * LROW];
  __shared__ __attribute__((aligned(16))) float Bs[NT * 32 * LROW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = tid & 7;        // 16-byte chunk inside the 32-float k chunk
  const int r0 = tid >> 3;      // 0..31
  const int M = p.B * p.H * p.W;
  const int HW = p.H * p.W;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * (NT * 32);
  const int nck = (p.Cin + BK - 1) / BK;
  const int nit = p.taps * nck;
  const bool affine = p.scale != nullptr;

  // per-thread A rows
  int a_b[4], a_y[4], a_x[4];
  bool a_ok[4];
  size_t a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    bool ok = m < M;
    int mm = ok ? m : 0;
    int b = mm / HW, rem = mm - b * HW;
    int y = rem / p.W, x = rem - y * p.W;
    if (ok && p.pro_mask != MASK_NONE) {
      bool lowp = is_low_freq(y, x, p.H, p.W, p.bs, p.low);
      ok = (p.pro_mask == MASK_LOW) ? lowp : !lowp;
    }
    a_b[i] = b; a_y[i] = y; a_x[i] = x; a_ok[i] = ok;
    a_off[i] = (size_t)mm * p.lda;
  }

  float4 a_reg[4], s_reg[4], h_reg[4], b_reg[NT];
  bool a_val[4];

  auto gload = [&](int it) {
    int tap = it / nck;
    int c = (it - tap * nck) * BK + 4 * q;
    int dy = 0, dx = 0;
    if (p.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
    bool cok = c < p.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = a_ok[i] && cok && (unsigned)(a_y[i] + dy) < (unsigned)p.H && (unsigned)(a_x[i] + dx) < (unsigned)p.W;
      a_val[i] = ok;
      a_reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        a_reg[i] = *reinterpret_cast<const float4*>(p.A + a_off[i] + (ptrdiff_t)(dy * p.W + dx) * p.lda + c);
        if (affine) {
          s_reg[i] = *reinterpret_cast<const float4*>(p.scale + (size_t)a_b[i] * p.Cin + c);
          h_reg[i] = *reinterpret_cast<const float4*>(p.shift + (size_t)a_b[i] * p.Cin + c);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      int n = n0 + r0 + 32 * i;
      b_reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < p.Cout && cok)
        b_reg[i] = *reinterpret_cast<const float4*>(p.Wt + ((size_t)tap * p.Cout + n) * p.Cin + c);
    }
  };

  auto lds_store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = a_reg[i];
      if (a_val[i]) {
        if (affine) {
          v.x = fmaf(v.x, s_reg[i].x, h_reg[i].x); v.y = fmaf(v.y, s_reg[i].y, h_reg[i].y);
          v.z = fmaf(v.z, s_reg[i].z, h_reg[i].z); v.w = fmaf(v.w, s_reg[i].w, h_reg[i].w);
        }
        if (p.pro_act != ACT_NONE) {
          v.x = act_apply(v.x, p.pro_act); v.y = act_apply(v.y, p.pro_act);
          v.z = act_apply(v.z, p.pro_act); v.w = act_apply(v.w, p.pro_act);
        }
      }
      *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LROW + 4 * q]) = v;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LROW + 4 * q]) = b_reg[i];
  };

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int lr = lane & 31;
  const int lh = lane >> 5;
  const float* a_rd = &As[(wave * 32 + lr) * LROW + 4 * lh];
  const float* b_rd = &Bs[lr * LROW + 4 * lh];

  gload(0);
  for (int it = 0; it < nit; ++it) {
    lds_store();
    __syncthreads();
    if (it + 1 < nit) gload(it + 1);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 a4 = *reinterpret_cast<const float4*>(a_rd + 8 * g);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float4 b4 = *reinterpret_cast<const float4*>(b_rd + nt * 32 * LROW + 8 * g);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  acc[nt][r]: row = (r&3) + 8*(r>>2) + 4*lh, col = lr   (C/D map of the 32x32 MFMA)
  const bool need_pos = p.tb || p.rowscale || p.epi_mask != MASK_NONE || p.store == STORE_NCHW;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    int m = m0 + row;
    if (m >= M) continue;
    int b = 0, y = 0, x = 0;
    if (need_pos) {
      b = m / HW;
      int rem = m - b * HW;
      y = rem / p.W;
      x = rem - y * p.W;
    }
    bool keep = true;
    if (p.epi_mask != MASK_NONE) {
      bool lowp = is_low_freq(y, x, p.H, p.W, p.bs, p.low);
      keep = (p.epi_mask == MASK_LOW) ? lowp : !lowp;
    }
    float rs = p.rowscale ? p.rowscale[b] : 1.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int n = n0 + nt * 32 + lr;
      if (n >= p.Cout) continue;
      float v = acc[nt][r];
      if (p.bias) v += p.bias[n];
      if (p.tb) v += p.tb[(size_t)b * p.ldtb + n];
      v = act_apply(v, p.epi_act);
      if (p.mul) v *= p.mul[(size_t)m * p.ldmul + n];
      if (p.rowscale) v *= rs;
      if (!keep) v = 0.f;
      if (p.res) v += p.res[(size_t)m * p.ldres + n];
      if (p.store == STORE_NHWC) p.out[(size_t)m * p.ldo + n] = v;
      else p.out[(((size_t)b * p.Cout + n) * p.H + y) * p.W + x] = v;
    }
  }
}


// ------------------------------------------------------------------------------------------------ bf16 variant
// Same tiling and fusions, operands rounded to bf16 on the way into LDS, v_mfma_f32_32x32x16_bf16 (fp32
// accumulate).  K chunk = 64 channels; LDS rows are 64 bf16 + 8 pad (144 B, conflict-free b128 reads); a lane
// reads 16 B = 8 consecutive k.  Weights come from the bf16 copy of the packed blob.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int BK16 = 64;
constexpr int LROW16 = BK16 * 2 + 16;   // bytes

template <int NT, int STAGES>
__device__ __forceinline__ void igemm_bf16_body(const IGemm& p, const int bx, const int by, const int nsplit, const int sk) {
  // STAGES == 2: two LDS stages: chunk i is consumed by the MFMAs while chunk i+1 is written and chunk i+2 is in flight
  // from global memory into registers -> one barrier per chunk and a full chunk of latency cover.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int A_BYTES = BM * LROW16, B_BYTES = NT * 32 * LROW16, STAGE = A_BYTES + B_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = tid & 7;        // 8-channel group inside the 64-channel chunk
  const int r0 = tid >> 3;
  const int M = p.B * p.H * p.W;
  const int HW = p.H * p.W;
  const int m0 = bx * BM;
  const int n0 = by * (NT * 32);
  const int nck = (p.Cin + BK16 - 1) / BK16;
  const int nit_all = p.taps * nck;
  // split-K: this workgroup owns iterations [it0, it1) of the (tap, chunk) space
  const int it0 = (int)((long long)nit_all * sk / nsplit), it1 = (int)((long long)nit_all * (sk + 1) / nsplit);
  const int nit = it1 - it0;
  const bool affine = p.scale != nullptr;
  const __bf16* Wb = reinterpret_cast<const __bf16*>(p.Wt16);

  int a_b[4], a_y[4], a_x[4];
  bool a_ok[4];
  size_t a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    bool ok = m < M;
    int mm = ok ? m : 0;
    int b = mm / HW, rem = mm - b * HW;
    int y = rem / p.W, x = rem - y * p.W;
    if (ok && p.pro_mask != MASK_NONE) {
      bool lowp = is_low_freq(y, x, p.H, p.W, p.bs, p.low);
      ok = (p.pro_mask == MASK_LOW) ? lowp : !lowp;
    }
    a_b[i] = b; a_y[i] = y; a_x[i] = x; a_ok[i] = ok;
    a_off[i] = (size_t)mm * p.lda;
  }

  float4 a_lo[4], a_hi[4];
  float4 sc_lo, sc_hi, sh_lo, sh_hi;      // GroupNorm scale/shift of this thread's channel group, image a_b[0]
  bool a_val[4], c_lo = false, c_hi = false;
  int cur_c = 0;
  bf16x8 b_reg[NT];

  auto gload = [&](int itl) {
    const int it = it0 + itl;
    int tap = it / nck;
    int c = (it - tap * nck) * BK16 + 8 * q;
    cur_c = c;
    int dy = 0, dx = 0;
    if (p.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
    c_lo = c < p.Cin; c_hi = c + 4 < p.Cin;           // Cin is a multiple of 4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = a_ok[i] && c_lo && (unsigned)(a_y[i] + dy) < (unsigned)p.H && (unsigned)(a_x[i] + dx) < (unsigned)p.W;
      a_val[i] = ok;
      a_lo[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      a_hi[i] = a_lo[i];
      if (ok && !(p.dbg & 1)) {
        const float* src = p.A + a_off[i] + (ptrdiff_t)(dy * p.W + dx) * p.lda + c;
        a_lo[i] = *reinterpret_cast<const float4*>(src);
        if (c_hi) a_hi[i] = *reinterpret_cast<const float4*>(src + 4);
      }
    }
    if (affine && c_lo) {
      const float* sp = p.scale + (size_t)a_b[0] * p.Cin + c;
      const float* hp = p.shift + (size_t)a_b[0] * p.Cin + c;
      sc_lo = *reinterpret_cast<const float4*>(sp); sh_lo = *reinterpret_cast<const float4*>(hp);
      if (c_hi) { sc_hi = *reinterpret_cast<const float4*>(sp + 4); sh_hi = *reinterpret_cast<const float4*>(hp + 4); }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      int n = n0 + r0 + 32 * i;
#pragma unroll
      for (int j = 0; j < 8; ++j) b_reg[i][j] = (__bf16)0.f;
      if (n < p.Cout && c_lo && !(p.dbg & 2)) {
        const __bf16* src = Wb + ((size_t)tap * p.Cout + n) * p.Cin + c;
        if (c_hi) b_reg[i] = *reinterpret_cast<const bf16x8*>(src);
        else {
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
          bf16x4 t4 = *reinterpret_cast<const bf16x4*>(src);
          b_reg[i][0] = t4[0]; b_reg[i][1] = t4[1]; b_reg[i][2] = t4[2]; b_reg[i][3] = t4[3];
        }
      }
    }
  };

  auto xform = [&](float v, float sc, float sh) {
    if (affine) v = fmaf(v, sc, sh);
    if (p.pro_act != ACT_NONE) v = act_apply(v, p.pro_act);
    return v;
  };

  auto lds_store = [&](int buf) {
    if (p.dbg & 16) return;
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 o;
      if (a_val[i]) {
        float4 s0 = sc_lo, s1 = sc_hi, h0 = sh_lo, h1 = sh_hi;
        if (affine && a_b[i] != a_b[0]) {   // tile straddles two images (rare): fetch that image's coefficients
          const float* sp = p.scale + (size_t)a_b[i] * p.Cin + cur_c;
          const float* hp = p.shift + (size_t)a_b[i] * p.Cin + cur_c;
          s0 = *reinterpret_cast<const float4*>(sp); h0 = *reinterpret_cast<const float4*>(hp);
          if (c_hi) { s1 = *reinterpret_cast<const float4*>(sp + 4); h1 = *reinterpret_cast<const float4*>(hp + 4); }
        }
        o[0] = (__bf16)xform(a_lo[i].x, s0.x, h0.x); o[1] = (__bf16)xform(a_lo[i].y, s0.y, h0.y);
        o[2] = (__bf16)xform(a_lo[i].z, s0.z, h0.z); o[3] = (__bf16)xform(a_lo[i].w, s0.w, h0.w);
        if (c_hi) {
          o[4] = (__bf16)xform(a_hi[i].x, s1.x, h1.x); o[5] = (__bf16)xform(a_hi[i].y, s1.y, h1.y);
          o[6] = (__bf16)xform(a_hi[i].z, s1.z, h1.z); o[7] = (__bf16)xform(a_hi[i].w, s1.w, h1.w);
        } else {
          o[4] = o[5] = o[6] = o[7] = (__bf16)0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(&As[(r0 + 32 * i) * LROW16 + 16 * q]) = o;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) *reinterpret_cast<bf16x8*>(&Bs[(r0 + 32 * i) * LROW16 + 16 * q]) = b_reg[i];
  };

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int a_rd = (wave * 32 + lr) * LROW16 + 16 * lh;
  const int b_rd = A_BYTES + lr * LROW16 + 16 * lh;

  if (STAGES == 1) {
    // short-K layers (full-resolution 3x3 / 1x1): one LDS stage, half the LDS -> more workgroups per CU cover
    // the load latency instead of a deeper in-workgroup pipeline
    if (nit > 0) gload(0);
    for (int it = 0; it < nit; ++it) {
      lds_store(0);
      __syncthreads();
      if (it + 1 < nit) gload(it + 1);
#pragma unroll
      for (int ks = 0; ks < BK16 / 16; ++ks) {
        bf16x8 a8 = *reinterpret_cast<const bf16x8*>(smem + a_rd + 32 * ks);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf16x8 b8 = *reinterpret_cast<const bf16x8*>(smem + b_rd + nt * 32 * LROW16 + 32 * ks);
          if (!(p.dbg & 4)) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[nt], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  } else if (nit > 0) {
    gload(0);
    lds_store(0);
    if (nit > 1) gload(1);
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      const unsigned char* st = smem + (it & 1) * STAGE;
#pragma unroll
      for (int ks = 0; ks < BK16 / 16; ++ks) {
        bf16x8 a8 = *reinterpret_cast<const bf16x8*>(st + a_rd + 32 * ks);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf16x8 b8 = *reinterpret_cast<const bf16x8*>(st + b_rd + nt * 32 * LROW16 + 32 * ks);
          if (!(p.dbg & 4)) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[nt], 0, 0, 0);
        }
      }
      if (it + 1 < nit) {
        lds_store((it + 1) & 1);          // stage (it+1)&1 was last read in iteration it-1 (fenced by its barrier)
        if (it + 2 < nit) gload(it + 2);
        __syncthreads();
      }
    }
  }

  const bool need_pos = p.tb || p.rowscale || p.epi_mask != MASK_NONE || p.store == STORE_NCHW;
  const bool atomic = nsplit > 1;
  const bool lead = sk == 0;

  if (!atomic && p.store == STORE_NHWC && (p.Cout & 3) == 0) {
    // ---- coalesced epilogue: accumulators -> wave-private LDS tile -> 16-byte row-contiguous global accesses
    // (4-byte-per-lane stores reach ~1 TB/s on this chip; full-line dwordx4 rows reach the HBM rate)
    constexpr int BN = NT * 32, BNP = BN + 4;
    __syncthreads();                                     // all waves are done with the staging buffers
    float* ct = reinterpret_cast<float*>(smem) + wave * 32 * BNP;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ct[((r & 3) + 8 * (r >> 2) + 4 * lh) * BNP + nt * 32 + lr] = acc[nt][r];
    // wave-private region, same wave reads it back: LDS ops of one wave are ordered
    constexpr int L4 = BN / 4;            // float4 per row
    constexpr int RPP = 64 / L4;          // rows per pass
    const int c4 = lane % L4, rsub = lane / L4;
    const int n = n0 + 4 * c4;
    if ((p.dbg & 8) && acc[0][0] != 12345.f) return;
    if (n < p.Cout) {
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && (!atomic || lead)) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll 4
      for (int pass = 0; pass < 32 / RPP; ++pass) {
        const int row = pass * RPP + rsub;
        const int m = m0 + wave * 32 + row;
        if (m >= M) continue;
        float4 v = *reinterpret_cast<const float4*>(ct + row * BNP + 4 * c4);
        int b = 0, y = 0, x = 0;
        if (need_pos) {
          b = m / HW;
          int rem = m - b * HW;
          y = rem / p.W;
          x = rem - y * p.W;
        }
        float* op = p.out + (size_t)m * p.ldo + n;
        if (atomic) {
          if (lead) {
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (p.tb) { float4 t = *reinterpret_cast<const float4*>(p.tb + (size_t)b * p.ldtb + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (p.res && p.res != p.out) { float4 t = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
          }
          atomicAdd(op, v.x); atomicAdd(op + 1, v.y); atomicAdd(op + 2, v.z); atomicAdd(op + 3, v.w);
          continue;
        }
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (p.tb) { float4 t = *reinterpret_cast<const float4*>(p.tb + (size_t)b * p.ldtb + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (p.epi_act != ACT_NONE) { v.x = act_apply(v.x, p.epi_act); v.y = act_apply(v.y, p.epi_act); v.z = act_apply(v.z, p.epi_act); v.w = act_apply(v.w, p.epi_act); }
        if (p.mul) { float4 t = *reinterpret_cast<const float4*>(p.mul + (size_t)m * p.ldmul + n); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
        if (p.rowscale) { float rs = p.rowscale[b]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
        if (p.epi_mask != MASK_NONE) {
          bool lowp = is_low_freq(y, x, p.H, p.W, p.bs, p.low);
          bool keep = (p.epi_mask == MASK_LOW) ? lowp : !lowp;
          if (!keep) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (p.res) { float4 t = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        *reinterpret_cast<float4*>(op) = v;
      }
    }
    return;
  }

  // ---- scalar epilogue (NCHW store of the 3-channel head, odd channel counts)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    int m = m0 + row;
    if (m >= M) continue;
    int b = 0, y = 0, x = 0;
    if (need_pos) {
      b = m / HW;
      int rem = m - b * HW;
      y = rem / p.W;
      x = rem - y * p.W;
    }
    bool keep = true;
    if (p.epi_mask != MASK_NONE) {
      bool lowp = is_low_freq(y, x, p.H, p.W, p.bs, p.low);
      keep = (p.epi_mask == MASK_LOW) ? lowp : !lowp;
    }
    float rs = p.rowscale ? p.rowscale[b] : 1.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int n = n0 + nt * 32 + lr;
      if (n >= p.Cout) continue;
      float v = acc[nt][r];
      if (atomic) {
        // split-K partial sums: one 256-byte row segment per wave instruction (the shape float atomics like);
        // linear epilogue only (checked by the launcher): the lead split adds bias / time bias / residual
        if (lead) {
          if (p.bias) v += p.bias[n];
          if (p.tb) v += p.tb[(size_t)b * p.ldtb + n];
          if (p.res && p.res != p.out) v += p.res[(size_t)m * p.ldres + n];
        }
        atomicAdd(&p.out[(size_t)m * p.ldo + n], v);
        continue;
      }
      if (p.bias) v += p.bias[n];
      if (p.tb) v += p.tb[(size_t)b * p.ldtb + n];
      v = act_apply(v, p.epi_act);
      if (p.mul) v *= p.mul[(size_t)m * p.ldmul + n];
      if (p.rowscale) v *= rs;
      if (!keep) v = 0.f;
      if (p.res) v += p.res[(size_t)m * p.ldres + n];
      if (p.store == STORE_NHWC) p.out[(size_t)m * p.ldo + n] = v;
      else p.out[(((size_t)b * p.Cout + n) * p.H + y) * p.W + x] = v;
    }
  }
}

// zero an [M][C] slice of a strided tensor (split-K accumulation target)
template <int NT, int STAGES>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IGemm p) {
  igemm_bf16_body<NT, STAGES>(p, blockIdx.x, blockIdx.y, gridDim.z, blockIdx.z);
}

// Up to four independent small problems of the same (Cout, Cin, taps) in ONE launch (blockIdx.z selects the problem):
// the AVIF block's pooled gates are four S x S maps (S = 1, 2, 4, 8) with their own weights, each a latency-bound GEMM of a
// few rows -- 8 launches of ~30 us per block become 2.
struct IGemmGroup { IGemm p[4]; int gx[4]; };
template <int NT, int STAGES>
__global__ __launch_bounds__(256) void igemm_bf16_group_kernel(const IGemmGroup g) {
  const int i = blockIdx.z;
  if ((int)blockIdx.x >= g.gx[i]) return;
  igemm_bf16_body<NT, STAGES>(g.p[i], blockIdx.x, blockIdx.y, 1, 0);
}

__global__ void zero_rows_k(float* out, int ldo, int64_t M, int C4) {
  const int64_t total = M * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t m = i / C4;
    int c4 = (int)(i - m * C4);
    *reinterpret_cast<float4*>(out + m * ldo + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int NT>
void launch_bf16(const IGemm& p, int64_t M, hipStream_t stream) {
  constexpr int EPI = 4 * 32 * (NT * 32 + 4) * 4;                      // epilogue staging tile
  constexpr int LDS1 = (BM + NT * 32) * LROW16 > EPI ? (BM + NT * 32) * LROW16 : EPI;
  constexpr int LDS = 2 * (BM + NT * 32) * LROW16;
  static PerDeviceOnce attr;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_bf16_kernel<NT, 2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_bf16_kernel<NT, 1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS1));
  });
  const int gx = (int)cdiv64(M, BM), gy = cdiv(p.Cout, NT * 32);
  const int nit = p.taps * cdiv(p.Cin, BK16);
  // split K across workgroups when the output tiling alone cannot fill the 256 CUs and the epilogue is linear
  int nsplit = 1;
  const bool linear = p.epi_act == ACT_NONE && !p.mul && !p.rowscale && p.epi_mask == MASK_NONE &&
                      p.store == STORE_NHWC && (p.Cout % 4) == 0 && (!p.res || p.res != p.out || p.ldres == p.ldo);
  // (split-K with float atomics is off: this kernel only serves tiny maps now, and bf16 mode stays bit-reproducible;
  //  set DDPMIR_IGEMM_SPLITK=1 to get the old behaviour for experiments)
  const bool allow_split = exp_switch("DDPMIR_IGEMM_SPLITK");
  if (allow_split && linear && gx * gy < 256 && nit >= 8) {
    nsplit = std::min(std::min(16, nit / 4), cdiv(512, gx * gy));
    if (nsplit < 1) nsplit = 1;
  }
  if (nsplit > 1 && !(p.res && p.res == p.out))
    hipLaunchKernelGGL(zero_rows_k, dim3((unsigned)std::min<int64_t>(cdiv64(M * (p.Cout / 4), 256), 2048)), dim3(256), 0,
                       stream, p.out, p.ldo, M, p.Cout / 4);
  const int force = exp_value("DDPMIR_IGEMM_STAGES");
  const bool one = force ? force == 1 : (nit <= 18 && nsplit == 1);
  if (one) hipLaunchKernelGGL((igemm_bf16_kernel<NT, 1>), dim3(gx, gy, 1), dim3(256), LDS1, stream, p);
  else hipLaunchKernelGGL((igemm_bf16_kernel<NT, 2>), dim3(gx, gy, nsplit), dim3(256), LDS, stream, p);
}

}  // namespace

template <int NT>
void launch_bf16_group(const IGemm* ps, int n, hipStream_t stream) {
  constexpr int EPI = 4 * 32 * (NT * 32 + 4) * 4;
  constexpr int LDS1 = (BM + NT * 32) * LROW16 > EPI ? (BM + NT * 32) * LROW16 : EPI;
  constexpr int LDS = 2 * (BM + NT * 32) * LROW16;
  static PerDeviceOnce attr;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_bf16_group_kernel<NT, 2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_bf16_group_kernel<NT, 1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS1));
  });
  IGemmGroup g;
  int gxmax = 1;
  for (int i = 0; i < 4; ++i) {
    g.p[i] = ps[i < n ? i : 0];
    g.gx[i] = i < n ? (int)cdiv64((int64_t)ps[i].B * ps[i].H * ps[i].W, BM) : 0;
    gxmax = std::max(gxmax, g.gx[i]);
  }
  const int gy = cdiv(ps[0].Cout, NT * 32), nit = ps[0].taps * cdiv(ps[0].Cin, BK16);
  if (nit <= 18) hipLaunchKernelGGL((igemm_bf16_group_kernel<NT, 1>), dim3(gxmax, gy, n), dim3(256), LDS1, stream, g);
  else hipLaunchKernelGGL((igemm_bf16_group_kernel<NT, 2>), dim3(gxmax, gy, n), dim3(256), LDS, stream, g);
}

void launch_igemm_group(const IGemm* ps, int n, hipStream_t stream) {
  bool same = n >= 1 && n <= 4 && ps[0].Wt16 != nullptr;
  for (int i = 1; i < n && same; ++i)
    same = ps[i].Wt16 && ps[i].Cout == ps[0].Cout && ps[i].Cin == ps[0].Cin && ps[i].taps == ps[0].taps;
  if (!same || n == 1) {            // fp32 parity path, or nothing to group
    for (int i = 0; i < n; ++i) launch_igemm(ps[i], stream);
    return;
  }
  for (int i = 0; i < n; ++i) {
    if (ps[i].Cin % 4 != 0 || ps[i].lda % 4 != 0) throw std::runtime_error("igemm: Cin and lda must be multiples of 4");
    if (ps[i].pro_mask != MASK_NONE && ps[i].taps != 1) throw std::runtime_error("igemm: input mask needs taps == 1");
  }
  if (ps[0].Cout <= 32) launch_bf16_group<1>(ps, n, stream);
  else if (ps[0].Cout <= 64) launch_bf16_group<2>(ps, n, stream);
  else launch_bf16_group<4>(ps, n, stream);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_igemm(const IGemm& p, hipStream_t stream) {
  if (p.Cin % 4 != 0 || p.lda % 4 != 0) throw std::runtime_error("igemm: Cin and lda must be multiples of 4");
  if (p.taps != 1 && p.taps != 9) throw std::runtime_error("igemm: taps must be 1 or 9");
  if (p.pro_mask != MASK_NONE && p.taps != 1) throw std::runtime_error("igemm: input mask needs taps == 1");
  const int64_t M = (int64_t)p.B * p.H * p.W;
  if (M <= 0 || p.Cout <= 0) return;
  dim3 block(256);
  if (p.Wt16) {
    if (p.Cout <= 32) launch_bf16<1>(p, M, stream);
    else if (p.Cout <= 64) launch_bf16<2>(p, M, stream);
    else launch_bf16<4>(p, M, stream);
    DDPMIR_HIP_CHECK(hipGetLastError());
    return;
  }
  if (p.Cout <= 32) {
    dim3 grid((unsigned)cdiv64(M, BM), 1);
    hipLaunchKernelGGL(igemm_kernel<1>, grid, block, 0, stream, p);
  } else if (p.Cout <= 64) {
    dim3 grid((unsigned)cdiv64(M, BM), 1);
    hipLaunchKernelGGL(igemm_kernel<2>, grid, block, 0, stream, p);
  } else {
    dim3 grid((unsigned)cdiv64(M, BM), (unsigned)cdiv(p.Cout, 128));
    hipLaunchKernelGGL(igemm_kernel<4>, grid, block, 0, stream, p);
  }
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

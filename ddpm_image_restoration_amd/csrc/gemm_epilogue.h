// Shared epilogue of the LDS-DMA GEMM kernels (gemm16.hip: 128-row four-wave tiles; gemm256.hip: 256 x 256 eight-wave tile).
#pragma once

#include "kernels.h"

// IGemm::dbg (skip A loads / B loads / MFMAs / stores: the dissection runs of tools/gemm_inventory.py) only exists in
// -DDDPMIR_EXPERIMENTS builds; in the shipped kernel the tests fold to constants.
#ifdef DDPMIR_EXPERIMENTS
#define GEMM_DBG(p) ((p).dbg)
#else
#define GEMM_DBG(p) 0
#endif

namespace ddpmir {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {

// Epilogue of the DMA GEMM kernels: split-K partial store, or accumulators -> wave-private LDS tile -> fused elementwise work
// and row-contiguous stores (also the fused qkv packing and the GroupNorm statistics).  acc[nt][r] holds row
// (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32 rows, column nt * 32 + lr.
// mw0 = first row of the wave's 32-row tile, n0 = first column; ct = the wave's private LDS tile (epi_ctw<NT>() floats).
// SYNC: drain the wave's DMAs and pass a workgroup barrier first (every wave is done reading the stage buffers ct overlays);
// a kernel that calls the epilogue several times per wave does that itself once.
template <int NT>
constexpr int epi_ctw() {
  constexpr int HN = NT < 2 ? NT : 2, BNP = HN * 32 + 4;
  return (NT == 4 && 64 * 36 > 32 * BNP) ? 64 * 36 : 32 * BNP;   // floats per wave (the V^T pack pass stores its tile transposed)
}
template <int NT, bool SYNC = true>
__device__ __forceinline__ void gemm16_epilogue(const IGemm& p, f32x16 (&acc)[NT], float* ct, int mw0, int n0, int M, int HW,
                                                int nsplit, int sk) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 31, lh = lane >> 5;
  // ------------------------------------------------------------------------------------------ epilogue
  if ((GEMM_DBG(p) & 8) && acc[0][0] != 12345.f) return;
  const bool need_pos = p.tb || p.rowscale || p.epi_mask != MASK_NONE || p.sel_C;
  const bool atomic = nsplit > 1;
  if (atomic) {
    // split-K: raw partial sums to the workspace slice of this split (128-byte row segments per wave instruction); the
    // linear epilogue runs in splitk_reduce_k, which adds the slices in a fixed order
    float* ws = p.splitk_ws + (size_t)sk * M * p.Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mw0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= M) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 32 + lr;
        if (n < p.Cout) ws[(size_t)m * p.Cout + n] = acc[nt][r];
      }
    }
    return;
  }
  // coalesced: accumulators -> wave-private LDS tile (HN column tiles at a time) -> 16-byte row-contiguous accesses
  constexpr int HN = NT < 2 ? NT : 2, BNH = HN * 32, BNP = BNH + 4;
  if constexpr (SYNC) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // all waves are done reading the stage buffers
  }
  constexpr int L4 = BNH / 4, RPP = 64 / L4;
  const int c4 = lane % L4, rsub = lane / L4;
  __bf16* out16 = reinterpret_cast<__bf16*>(p.out16);
#pragma unroll
  for (int h = 0; h < NT / HN; ++h) {
    if constexpr (NT == 4) {
      if (p.pk_q) {   // fused qkv projection: this 32-token x 64-column tile is Q, K or V of one 32-key attention tile
        // `heads` heads of head_dim D = 8, 16 or 32 (C = heads * D, a multiple of 64); a 64-column pass holds 64/D heads of
        // NCH 16-deep chunks each (head_dim 8 is zero-padded to one 16-deep chunk: AttnGeom<8>::DP = 16)
        const int D = p.pk_D, nheads = p.pk_heads, C = nheads * D, NCH = D < 16 ? 1 : D >> 4, HPP = 64 / D, DPq = D < 16 ? 16 : D;
        const int mrow0 = mw0, nb = n0 + h * BNH;
        if (mrow0 >= M || nb >= 3 * C) continue;
        const int b = (int)p.fd_hw.div((unsigned)mrow0), tile = (mrow0 - b * HW) >> 5;
        const int region = nb / C, head0 = (nb - region * C) / D;      // 0 = Q, 1 = K, 2 = V
        const int kt_bytes = NCH * 1024, rv = D <= 16 ? D + 1 : 32, tile_bytes = kt_bytes + 4 * rv * 16;
        if (region < 2) {
          // Q / K: row-major tile [token][64 + 4], bias (and the softmax scale for Q) folded in on the way
          const float mulv = region == 0 ? p.pk_qscale : 1.0f;
#pragma unroll
          for (int q = 0; q < HN; ++q) {
            const float bcol = p.bias[nb + q * 32 + lr];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ct[((r & 3) + 8 * (r >> 2) + 4 * lh) * BNP + q * 32 + lr] = (acc[h * HN + q][r] + bcol) * mulv;
          }
          // lane (hh = lane >> 5, token = lane & 31) owns components 16c + 8hh .. +7 of chunk c of every head
          const int tok = lane & 31, hh = lane >> 5;
          for (int hp = 0; hp < HPP; ++hp) {
            const size_t bh_tile = (size_t)(b * nheads + head0 + hp) * p.pk_nt + tile;
            float nk = 0.f;
            for (int c = 0; c < NCH; ++c) {
              const float* src = ct + tok * BNP + hp * D + 16 * c + hh * 8;
              float4 a = make_float4(0.f, 0.f, 0.f, 0.f), e = a;
              if (16 * c + hh * 8 < D) { a = *reinterpret_cast<const float4*>(src); e = *reinterpret_cast<const float4*>(src + 4); }   // (head_dim 8: the upper half-chunk is padding)
              bf16x8 v;
              v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
              v[4] = (__bf16)e.x; v[5] = (__bf16)e.y; v[6] = (__bf16)e.z; v[7] = (__bf16)e.w;
              if (region == 0) {         // Q[bh][token][DP]
                *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.pk_q) + (bh_tile * 32 + tok) * DPq + 16 * c + hh * 8) = v;
              } else {                   // K image: chunk c, lane l = hh*32 + key at byte c*1024 + l*16
                *reinterpret_cast<bf16x8*>(reinterpret_cast<unsigned char*>(p.pk_kv) + bh_tile * tile_bytes + c * 1024 + lane * 16) = v;
              }
#pragma unroll
              for (int j2 = 0; j2 < 8; ++j2) { const float f = (float)v[j2]; nk = fmaf(f, f, nk); }   // |q|^2 or |k|^2 of this token
            }
            if (region == 1) {           // tile max of |k| on the rounded values (bound of any score of this tile)
              nk += __shfl_xor(nk, 32);
#pragma unroll
              for (int off = 16; off > 0; off >>= 1) nk = fmaxf(nk, __shfl_xor(nk, off));
              if (lane == 0) p.pk_kmax[bh_tile] = sqrtf(nk) * 1.002f;
            } else {                     // tile max of |q|^2 (scaled, rounded): attn_pair_bounds_k folds the tiles; NaN -> +inf
              nk += __shfl_xor(nk, 32);
              if (!(nk == nk)) nk = INFINITY;
#pragma unroll
              for (int off = 16; off > 0; off >>= 1) nk = fmaxf(nk, __shfl_xor(nk, off));
              if (lane == 0) p.pk_qmax[bh_tile] = nk;
            }
          }
        } else {
          // V: TRANSPOSED tile [column][32 tokens + 4] so the 4 consecutive keys of a V^T slot group are one float4
          constexpr int TP = 36;
#pragma unroll
          for (int q = 0; q < HN; ++q) {
            const float bcol = p.bias[nb + q * 32 + lr];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
              *reinterpret_cast<float4*>(ct + (q * 32 + lr) * TP + 8 * g4 + 4 * lh) =
                  make_float4(acc[h * HN + q][4 * g4] + bcol, acc[h * HN + q][4 * g4 + 1] + bcol,
                              acc[h * HN + q][4 * g4 + 2] + bcol, acc[h * HN + q][4 * g4 + 3] + bcol);
          }
          // V^T image: slab (s, hh), row d, slot j = key (j&3) + 8*(2s + (j>>2)) + 4hh; 64 columns x 4 slabs = 4 items per lane
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int item = it * 64 + lane, col = item & 63, slab = item >> 6;      // col = head-in-pass * D + row
            const int hp = col / D, row = col - hp * D, sI = slab >> 1, hh = slab & 1;
            const float* src = ct + col * TP + 16 * sI + 4 * hh;
            const float4 a = *reinterpret_cast<const float4*>(src), e = *reinterpret_cast<const float4*>(src + 8);
            bf16x8 v;
            v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
            v[4] = (__bf16)e.x; v[5] = (__bf16)e.y; v[6] = (__bf16)e.z; v[7] = (__bf16)e.w;
            unsigned char* vb = reinterpret_cast<unsigned char*>(p.pk_kv) + ((size_t)(b * nheads + head0 + hp) * p.pk_nt + tile) * tile_bytes + kt_bytes;
            *reinterpret_cast<bf16x8*>(vb + (size_t)slab * rv * 16 + row * 16) = v;
          }
          if (D <= 16 && lane < 4 * HPP) {   // the ones row (softmax denominator): row D of each (head, slab) of this pass
            bf16x8 one;
#pragma unroll
            for (int j2 = 0; j2 < 8; ++j2) one[j2] = (__bf16)1.0f;
            const int hd = head0 + (lane >> 2), slab = lane & 3;
            unsigned char* vb = reinterpret_cast<unsigned char*>(p.pk_kv) + ((size_t)(b * nheads + hd) * p.pk_nt + tile) * tile_bytes + kt_bytes;
            *reinterpret_cast<bf16x8*>(vb + (size_t)slab * rv * 16 + D * 16) = one;
          }
        }
        continue;
      }
    }
#pragma unroll
    for (int q = 0; q < HN; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) ct[((r & 3) + 8 * (r >> 2) + 4 * lh) * BNP + q * 32 + lr] = acc[h * HN + q][r];
    const int n = n0 + h * BNH + 4 * c4;
    if (n >= p.Cout) continue;                          // Cout % 4 == 0: a float4 is all in or all out
    const bool hi_cols = p.sel_C && n >= p.sel_C;       // sel_C % 64 == 0: a 64-column pass is all low or all high
    const int no = hi_cols ? n - p.sel_C : n;           // column of mul / res / out
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
    float gs1 = 0.f, gs2 = 0.f;                         // GroupNorm statistics of this lane's 4 columns over its rows
#pragma unroll 4
    for (int pass = 0; pass < 32 / RPP; ++pass) {
      const int row = pass * RPP + rsub;
      const int m = mw0 + row;
      if (m >= M) continue;
      float4 v = *reinterpret_cast<const float4*>(ct + row * BNP + 4 * c4);
      int b = 0, y = 0, x = 0;
      if (need_pos) {
        b = (int)p.fd_hw.div((unsigned)m);
        if (p.epi_mask != MASK_NONE || p.sel_C) {       // only the frequency masks need the pixel's position inside its image
          const int rem = m - b * HW;
          y = (int)p.fd_w.div((unsigned)rem);
          x = rem - y * p.W;
        }
      }
      if (p.sel_C && is_low_freq(y, x, p.H, p.W, p.bs, p.low) == hi_cols) continue;   // the other gate owns this pixel
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (p.tb) { float4 t = *reinterpret_cast<const float4*>(p.tb + (size_t)b * p.ldtb + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      if (p.epi_act != ACT_NONE) { v.x = act16(v.x, p.epi_act); v.y = act16(v.y, p.epi_act); v.z = act16(v.z, p.epi_act); v.w = act16(v.w, p.epi_act); }
      if (p.mul) { float4 t = *reinterpret_cast<const float4*>(p.mul + (size_t)m * p.ldmul + no); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
      if (p.rowscale && (!p.sel_C || hi_cols)) { const float rs = p.rowscale[b]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
      if (p.epi_mask != MASK_NONE) {
        const bool lowp = is_low_freq(y, x, p.H, p.W, p.bs, p.low);
        const bool keep = (p.epi_mask == MASK_LOW) ? lowp : !lowp;
        if (!keep) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (p.res) { float4 t = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + no); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      else if (p.res16) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.res16) + (size_t)m * p.ldres + no);
        v.x += (float)t[0]; v.y += (float)t[1]; v.z += (float)t[2]; v.w += (float)t[3];
      }
      if (p.gn_sums) {
        // the sums are those of the values the consumer will normalise: where the result only exists as bf16, of the rounded
        // values (as head_mix16_k and stream_k do)
        float4 u = v;
        if (out16 && !p.out) { u.x = (float)(__bf16)v.x; u.y = (float)(__bf16)v.y; u.z = (float)(__bf16)v.z; u.w = (float)(__bf16)v.w; }
        gs1 += (u.x + u.y) + (u.z + u.w);
        gs2 += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
      }
      if (p.out) *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldo + no) = v;
      if (out16) {
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(out16 + (size_t)m * p.ldo16 + no) = o;
      }
      if (p.out8) *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(p.out8) + (size_t)m * p.ldo8 + no) = pack_e4m3x4(v);
    }
    if (p.gn_sums) {
      // the wave's 32 rows lie in one image (HW % 32 == 0); fold rows (lanes L4 apart), then the lanes of one group.  Every
      // (32-row wave tile, group) has exactly ONE writer -- a group lies inside one 64-column pass (cpg <= 64) -- which stores
      // its partial as a double at a slot of its own: no atomics, no clearing pass, and gn_finalize folds the slots of a
      // (sample, group) in a fixed order, so the statistics are bit-reproducible.
      const int cpg = p.Cout / p.gn_groups, lg = cpg / 4;             // lanes per group within a row segment
#pragma unroll
      for (int off = L4; off < 64; off <<= 1) { gs1 += __shfl_xor(gs1, off); gs2 += __shfl_xor(gs2, off); }
      const int span = lg < L4 ? lg : L4;
      for (int off = 1; off < span; off <<= 1) { gs1 += __shfl_xor(gs1, off); gs2 += __shfl_xor(gs2, off); }
      const int mw = mw0;
      if (rsub == 0 && (c4 % span) == 0 && mw < M) {
        const int b = (int)p.fd_hw.div((unsigned)mw), wt = (mw - b * HW) >> 5;
        double* dst = p.gn_sums + (((size_t)b * (HW >> 5) + wt) * p.gn_groups + n / cpg) * 2;
        dst[0] = (double)gs1;
        dst[1] = (double)gs2;
      }
    }
  }
}


// Register-direct epilogue of the 64-column kernels for the plain option sets the full-resolution blocks use (conv1: bias +
// time bias -> bf16 + GroupNorm partials; conv2: bias -> bf16 (+ fp32); conv_out: bias -> fp32 (+ partials)).  The accumulators
// go to memory in their own layout -- lane = output column, register = row -- with 4-byte (fp32) / 2-byte (bf16) stores: 32
// consecutive columns of a row are one 128- / 64-byte piece and the wave's following stores complete the lines in L2
// (tools/micro/store_pattern.hip: 6.7 TB/s, the same as whole 128-byte rows).  No LDS tile, no barrier, and -- unlike the first
// attempt of round 4, which kept every option of gemm16_epilogue behind per-element branches and ran 100-150 us slower per
// launch (profiles/r04_epilogue_direct_ab.txt) -- no branch inside the row loop: the option set is a template argument, whole
// 32-row wave tiles only (HW % 32 == 0), whole column tiles only.  Same arithmetic in the same order as gemm16_epilogue, so the
// stored values are bit-identical; the fused statistics differ in summation order only.
// mode bits: 1 = fp32 output, 2 = bf16 output, 4 = GroupNorm partials.  0 = not eligible (LDS epilogue).
inline int epilogue_direct_mode(const IGemm& p, int nt_cols) {
  if (p.pk_q || p.sel_C || p.epi_mask != MASK_NONE || p.out8 || p.mul || p.rowscale || p.res || p.res16 || p.epi_act != ACT_NONE ||
      p.store != STORE_NHWC || !p.bias || (!p.out && !p.out16) || ((p.H * p.W) & 31) != 0 || (p.Cout % nt_cols) != 0)
    return 0;
  if (p.gn_sums) {
    const int cpg = p.gn_groups > 0 ? p.Cout / p.gn_groups : 0;
    if (cpg < 4 || (cpg & (cpg - 1)) != 0 || cpg > 64 || (cpg == 64 && nt_cols < 64)) return 0;
  }
  // (measured, tools/epilogue_ab.py: with BOTH copies to store the direct form wins for the 3x3 convs -- 8 -> 64: 238 -> 169 us,
  //  64 -> 64: 359 -> 334 -- and loses for 1x1 launches -- 64 -> 64 at 512 x 512: 182 -> 232 us: those keep the LDS epilogue)
  if (p.out && p.out16 && p.taps != 9) return 0;
  return (p.out ? 1 : 0) | (p.out16 ? 2 : 0) | (p.gn_sums ? 4 : 0);
}

template <int NT, bool OUT32, bool OUT16, bool GN>
__device__ __forceinline__ void gemm16_epilogue_direct(const IGemm& p, f32x16 (&acc)[NT], int mw0, int n0, int M, int HW) {
  if (mw0 >= M) return;                                   // (M % 32 == 0: a wave tile is all rows or none)
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  const int b = (p.tb || GN) ? (int)p.fd_hw.div((unsigned)mw0) : 0;
  float gs1[NT], gs2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + nt * 32 + lr;
    const float bias = p.bias[n];
    const float tbv = p.tb ? p.tb[(size_t)b * p.ldtb + n] : 0.f;
    float* o32 = OUT32 ? p.out + (size_t)(mw0 + 4 * lh) * p.ldo + n : nullptr;
    __bf16* o16 = OUT16 ? reinterpret_cast<__bf16*>(p.out16) + (size_t)(mw0 + 4 * lh) * p.ldo16 + n : nullptr;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2);               // row of the half-wave's 16: compile-time constant
      float v = acc[nt][r] + bias;
      if (p.tb) v += tbv;                                 // (wave-uniform; same order as gemm16_epilogue: (acc + bias) + tb)
      if (OUT32) o32[(size_t)k * p.ldo] = v;
      if (OUT16) o16[(size_t)k * p.ldo16] = (__bf16)v;
      if (GN) {
        const float u = (OUT16 && !OUT32) ? (float)(__bf16)v : v;   // statistics of what the consumer will read
        s1 += u;
        s2 += u * u;
      }
    }
    gs1[nt] = s1; gs2[nt] = s2;
  }
  if (GN) {
    // one writer per (32-row wave tile, group): fold the two row halves (lanes 32 apart), then the lanes of a group
    const int cpg = p.Cout / p.gn_groups, wt = (mw0 - b * HW) >> 5;
    if (NT >= 2 && cpg == 64) {
#pragma unroll
      for (int nt = 0; nt + 1 < NT; nt += 2) { gs1[nt] += gs1[nt + 1]; gs2[nt] += gs2[nt + 1]; }
    }
    const int span = cpg < 32 ? cpg : 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (NT >= 2 && cpg == 64 && (nt & 1)) continue;
      float s1 = gs1[nt], s2 = gs2[nt];
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      for (int off = 1; off < span; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
      const int n = n0 + nt * 32 + lr;
      if (lh == 0 && (lr & (span - 1)) == 0) {
        double* dst = p.gn_sums + (((size_t)b * (HW >> 5) + wt) * p.gn_groups + n / cpg) * 2;
        dst[0] = (double)s1;
        dst[1] = (double)s2;
      }
    }
  }
}

// dispatch on the mode computed by the launcher (wave-uniform switch, once per tile)
template <int NT>
__device__ __forceinline__ void gemm16_epilogue_direct_any(const IGemm& p, f32x16 (&acc)[NT], int mw0, int n0, int M, int HW) {
  switch (p.epi_direct) {
    case 1: gemm16_epilogue_direct<NT, true, false, false>(p, acc, mw0, n0, M, HW); break;
    case 2: gemm16_epilogue_direct<NT, false, true, false>(p, acc, mw0, n0, M, HW); break;
    case 3: gemm16_epilogue_direct<NT, true, true, false>(p, acc, mw0, n0, M, HW); break;
    case 5: gemm16_epilogue_direct<NT, true, false, true>(p, acc, mw0, n0, M, HW); break;
    case 6: gemm16_epilogue_direct<NT, false, true, true>(p, acc, mw0, n0, M, HW); break;
    default: gemm16_epilogue_direct<NT, true, true, true>(p, acc, mw0, n0, M, HW); break;
  }
}

}  // namespace
}  // namespace ddpmir

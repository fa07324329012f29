// bf16 matrix-core self-attention for the restoration U-Net (reference: nn.MultiheadAttention,
// webp_inference.py:295,317-321; 95 % of the FLOPs at 256^2 sit in the two full-resolution blocks with
// N = 65 536 tokens and head_dim 16, where the kernel is bound by v_exp_f32 issue, not by MFMA).
//
// Kernels:
//  pack_qkv_bf16<D>   fp32 qkv [B][N][3C]  ->  bf16 operand images laid out exactly as the wave reads them
//                     (head_dim 8 / 16 / 32 with C a multiple of 64: written by the in_proj GEMM's epilogue instead, gemm16.hip):
//      Qp [bh][Npad][DP]                       q * log2(e)/sqrt(d), d zero-padded to DP (multiple of 16)
//      Kp [bh][tile][c][h][32 keys][8]         one 16-deep k chunk c, lane half h: lane l reads 16 B at l*16
//      Vp [bh][tile][dt][s][h][rows][8]        V transposed AND key-permuted: slot j of lane half h at MFMA step s
//                                              is key (j&3) + 8*(2s + (j>>2)) + 4h of the tile -- the key the
//                                              S^T accumulator register 8s+j of that lane half holds.
//      For D <= 16 the tile carries an extra row of ones (zeros for padding keys), so the PV MFMA also
//      produces the softmax denominator in a spare accumulator row for free.
//  flash_bf16<D,QS>   one wave = QS x 32 query rows, everything keyed by the query on the lane:
//      S^T = K Q^T with the accumulator initialised to -m (running max) -> s - m comes out of the MFMA,
//      p = v_exp_f32(s - m), packed to bf16 in place and used AS IS as the B operand of O^T += V^T P^T.
//      The running max is only raised when a tile exceeds it by more than RESCALE_THR (deferred rescale;
//      fp32 accumulators and bf16's 8-bit exponent make 2^THR-sized p harmless), so the per-tile VALU work
//      is 16 v_exp + 8 v_max3 + 8 v_cvt_pk per 1024 scores.
//  flash_bf16_nm<D,QS,TS>  the same without a running max (and without its registers: three waves per SIMD) for the
//      (batch, head) pairs whose scores are provably bounded (attn_pair_bounds_k: max|q| * max|k| <= NM_BOUND in log2
//      units); head_dim <= 64.  The general kernel skips those pairs.
//  K/V stream through LDS in stages of several 32-key tiles, double buffered, filled by global_load_lds (the packed
//  images are a linear copy), laid out so every LDS read is a linear 1 KiB wave read (conflict-free; the one exception is the
//  V^T slab read at head_dim 16, 2-way on purpose: vt_row below).  Workgroups of one
//  (batch, head) are placed on one XCD so the 4 MiB of K/V they all re-read stay in that XCD's L2.
#include <cstdlib>
#include <stdexcept>

#include "kernels.h"

namespace ddpmir {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

template <int D>
struct AttnGeom {
  static constexpr int DP = D < 16 ? 16 : D;               // k-padded head dim
  static constexpr int KC = DP / 16;                       // 16-deep k chunks of QK^T
  static constexpr bool ONES = D <= 16;                    // denominator via a ones row of V
  static constexpr int DT = ONES ? 1 : D / 32;             // 32-row output tiles of O^T
  static constexpr int RV = ONES ? D + 1 : 32;             // stored rows per (dt,s,h) slab of V^T
  static constexpr int KT_BYTES = KC * 1024;
  static constexpr int VT_BYTES = DT * 2 * 2 * RV * 16;
  static constexpr int TILE_BYTES = KT_BYTES + VT_BYTES;
  // default tiles per LDS stage: ~16 KiB per buffer
  static constexpr int TS_DEFAULT = D <= 16 ? 8 : (D == 32 ? 8 : (D == 64 ? 4 : 2));
};

template <int D, int TS>
struct StageGeom {
  static constexpr int STAGE_BYTES = TS * AttnGeom<D>::TILE_BYTES;
  static constexpr int CHUNKS = STAGE_BYTES / 16;
  static constexpr int CPT = (CHUNKS + 255) / 256;         // 16-byte chunks per thread per stage
};

// key held by S^T accumulator register r of lane half h (C/D map of the 32x32 MFMA)
__device__ __forceinline__ int acc_key(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------ pack
template <int D>
__global__ __launch_bounds__(256) void pack_qkv_bf16(const float* __restrict__ qkv, __bf16* __restrict__ Qp,
                                                     unsigned char* __restrict__ KVp, float* __restrict__ Kmax,
                                                     int N, int NT, int C, float qscale) {
  using G = AttnGeom<D>;
  __shared__ float sk[32][D + 1];
  __shared__ float sv[32][D + 1];
  const int tile = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int heads = gridDim.y;
  const int bh = b * heads + head;
  const int tid = threadIdx.x;
  const float* base = qkv + (size_t)b * N * 3 * C + head * D;
  // stage K and V of the 32 tokens; write Q directly
  for (int i = tid; i < 32 * D; i += 256) {
    int tok = i / D, c = i - tok * D;
    int n = tile * 32 + tok;
    float kq = 0.f, vq = 0.f, qq = 0.f;
    if (n < N) {
      const float* rp = base + (size_t)n * 3 * C + c;
      qq = rp[0]; kq = rp[C]; vq = rp[2 * C];
    }
    sk[tok][c] = kq; sv[tok][c] = vq;
    Qp[((size_t)bh * NT * 32 + n) * G::DP + c] = (__bf16)(qq * qscale);
  }
  if (G::DP > D)
    for (int i = tid; i < 32 * (G::DP - D); i += 256) {
      int tok = i / (G::DP - D), c = D + i % (G::DP - D);
      Qp[((size_t)bh * NT * 32 + tile * 32 + tok) * G::DP + c] = (__bf16)0.f;
    }
  __syncthreads();
  if (tid < 64) {   // largest |k|_2 of the tile (on the bf16-rounded values): upper bound of any score q.k <= |q||k|
    float nk = 0.f;
    if (tid < 32)
      for (int c = 0; c < D; ++c) { float kv = (float)(__bf16)sk[tid][c]; nk = fmaf(kv, kv, nk); }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) nk = fmaxf(nk, __shfl_xor(nk, off));
    if (tid == 0) Kmax[(size_t)bh * NT + tile] = sqrtf(nk) * 1.002f;
  }
  unsigned char* tbase = KVp + ((size_t)bh * NT + tile) * G::TILE_BYTES;
  // K image: chunk (c, h, key) = comps 16c + 8h + j
  for (int i = tid; i < G::KC * 64; i += 256) {
    int c = i / 64, l = i - c * 64, h = l >> 5, key = l & 31;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int comp = 16 * c + 8 * h + j;
      v[j] = (__bf16)(comp < D ? sk[key][comp] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(tbase + c * 1024 + l * 16) = v;
  }
  // V^T image: slab (dt, s, h), row dd, slot j = key (j&3) + 8*(2s + (j>>2)) + 4h
  unsigned char* vbase = tbase + G::KT_BYTES;
  for (int i = tid; i < G::DT * 4 * G::RV; i += 256) {
    int slab = i / G::RV, row = i - slab * G::RV;
    int dt = slab >> 2, s = (slab >> 1) & 1, h = slab & 1;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int key = acc_key(8 * s + j, h);
      float val;
      if (G::ONES && row == D) val = (tile * 32 + key < N) ? 1.0f : 0.0f;
      else val = sv[key][dt * 32 + row];
      v[j] = (__bf16)val;
    }
    *reinterpret_cast<bf16x8*>(vbase + (size_t)slab * G::RV * 16 + row * 16) = v;
  }
}

// ------------------------------------------------------------------------------------------------ flash
constexpr float RESCALE_THR = 20.0f;   // log2 units
// (the score bound below which the softmax needs no running max, flash_bf16_nm further down, is AttnOpts::nm_bound: 80 in log2 units)
struct PairBounds { unsigned qmax2, kmax; };   // fp32 bit patterns of max |q|^2 and max |k|
// Routing is per (pair, ROUTING BLOCK of queries): a block of RB = 128 * NMQS consecutive queries (one workgroup of the no-max
// kernel) runs without a running max when  max|q| of ITS queries * max|k| of the pair <= nm_bound  -- one outlier query token
// sends its own block to the general kernel, not the pair's 65 536 queries.  blkq[bh][rb] = max |q|^2 of the block.
__device__ __forceinline__ bool block_is_bounded(const PairBounds* bounds, const float* blkq, int bh, int nrb, int rb, float nm_bound) {
  const float q2 = blkq[(size_t)bh * nrb + rb], km = __uint_as_float(bounds[bh].kmax);
  return sqrtf(q2) * 1.002f * km <= nm_bound;
}

// Row of the V^T slab a lane reads as its A-operand row.  Rows 0 .. D-1 are V^T, row D the ones row (head_dim <= 16): lanes
// D .. 31 ALL read the ones row (only A row D must be it; rows D+1 .. 31 of O^T are never read).
// head_dim 16: ds_read_b128 is served in the lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32) and the ones row sits one
// 256-byte bank row above row 0, so the first group holds rows 0-3 and row 16 on the same banks: 2-way, SQ_LDS_BANK_CONFLICT =
// 25 % of this kernel's LDS cycles.  Round 3 removed that count by letting lanes 17-31 re-read rows 1-15 (every group 16 distinct
// slots: conflicts 67.1 M -> 0) -- and a same-device A/B of the two builds (tools/attn_lib_ab.py, profiles/r03_attn_vrow_ab.txt)
// showed the conflict-free pattern 2.2 % SLOWER per level-0 launch (25.61 vs 25.07 ms): fifteen lanes on one broadcast row cost
// the LDS less than fifteen more distinct rows, and LDS is 8-10 % busy here anyway.  The counter is explained; the pattern stays.
template <int D, bool ONES>
__device__ __forceinline__ int vt_row(int lr) {
  if (!ONES) return lr;
  return lr < D ? lr : D;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// acc += sum of the 16 packed bf16 values of pb (two independent chains)
__device__ __forceinline__ void sum_packed(const bf16x8 (&pb)[2], float2& acc) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 one2;
  one2[0] = (__bf16)1.0f; one2[1] = (__bf16)1.0f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x2 v;
      v[0] = pb[h][2 * j]; v[1] = pb[h][2 * j + 1];
      if (j & 1) acc.y = __builtin_amdgcn_fdot2_f32_bf16(v, one2, acc.y, false);
      else acc.x = __builtin_amdgcn_fdot2_f32_bf16(v, one2, acc.x, false);
    }
}

template <int D, int QS, int TS, int WPS = 0, int DMA = 1, int DEN = 0>
__global__ __launch_bounds__(256, (WPS > 0 ? WPS : 1)) void flash_bf16(const __bf16* __restrict__ Qp, const unsigned char* __restrict__ KVp,
                                                  const float* __restrict__ Kmax, float* __restrict__ out, int ldo,
                                                  int N, int NT, int heads, int nqb, int npairs,
                                                  __bf16* __restrict__ out16, const PairBounds* __restrict__ skip,
                                                  const float* __restrict__ blkq, int nrb, int rbq, float nm_bound,
                                                  unsigned* __restrict__ taken) {
  using G = AttnGeom<D>;
  using SG = StageGeom<D, TS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 stage buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;

  // XCD-aware placement: workgroup ids are dealt round-robin over the 8 XCDs, so give every (batch, head)
  // pair to one residue class -- all query blocks of a pair then share that XCD's L2 for K/V.
  int wg = blockIdx.x, bh, qb;
  if ((npairs & 7) == 0) {
    int xcd = wg & 7, idx = wg >> 3;
    bh = (idx / nqb) * 8 + xcd;
    qb = idx - (idx / nqb) * nqb;
  } else {
    bh = wg / nqb;
    qb = wg - bh * nqb;
  }
  // bounded scores: this query block ran in the no-max kernel (flash_bf16_nm); rbq = general workgroups per routing block
  if (skip && block_is_bounded(skip, blkq, bh, nrb, qb / rbq, nm_bound)) return;
  if (skip && taken && tid == 0) atomicAdd(taken, 1u);   // profiling: workgroups that needed the running max
  const int b = bh / heads, head = bh - b * heads;
  const int q0 = qb * (128 * QS) + wave * (32 * QS);

  // Q fragments: lane (q = lr, half lh) holds comps 16c + 8lh + j
  bf16x8 qf[QS][G::KC];
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    int qr = q0 + 32 * u + lr;
    if (qr >= NT * 32) qr = NT * 32 - 1;
    const __bf16* qp = Qp + ((size_t)bh * NT * 32 + qr) * G::DP + 8 * lh;
#pragma unroll
    for (int c = 0; c < G::KC; ++c) qf[u][c] = *reinterpret_cast<const bf16x8*>(qp + 16 * c);
  }

  // |q|_2 per query row (both lane halves hold half of the components)
  float qn[QS], m_run[QS];
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < G::KC; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { float v = (float)qf[u][c][j]; a = fmaf(v, v, a); }
    a += __shfl_xor(a, 32);
    qn[u] = sqrtf(a) * 1.002f;
    m_run[u] = 0.f;
  }

  f32x16 o[QS][G::DT], negm[QS];
  float lsum[QS];
  // head_dim >= 32 has no spare accumulator row for the ones trick: the softmax denominator comes from one more
  // MFMA per k-step with an all-ones A operand (8 VALU issue cycles instead of a 16-add chain per tile)
  f32x16 osum[QS];
  float2 lacc[QS];
#pragma unroll
  for (int u = 0; u < QS; ++u) lacc[u] = make_float2(0.f, 0.f);
  bf16x8 ones8;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones8[j] = (__bf16)1.0f;
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    lsum[u] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) osum[u][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[u][r] = 0.f;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][t][r] = 0.f;
  }

  const unsigned char* gkv = KVp + (size_t)bh * NT * G::TILE_BYTES;
  const int nstages = (NT + TS - 1) / TS;
  const size_t total_bytes = (size_t)NT * G::TILE_BYTES;
  const int vrow = vt_row<D, G::ONES>(lr);               // row of the V^T slab this lane reads

  float* kmax_lds = reinterpret_cast<float*>(lds + 2 * SG::STAGE_BYTES);   // [2][TS]
  const float* gkmax = Kmax + (size_t)bh * NT;
  float4 stg[SG::CPT];
  float stg_km = 0.f;
  auto gload = [&](int stage) {
    if (tid < TS) stg_km = (stage * TS + tid < NT) ? gkmax[stage * TS + tid] : 0.f;
    const unsigned char* src = gkv + (size_t)stage * SG::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < SG::CPT; ++i) {
      size_t off = ((size_t)i * 256 + tid) * 16;
      stg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((size_t)stage * SG::STAGE_BYTES + off < total_bytes && i * 256 + tid < SG::CHUNKS)
        stg[i] = *reinterpret_cast<const float4*>(src + off);
    }
  };
  auto lstore = [&](int buf) {
    if (tid < TS) kmax_lds[buf * TS + tid] = stg_km;
    unsigned char* dst = lds + buf * SG::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < SG::CPT; ++i)
      if (i * 256 + tid < SG::CHUNKS) *reinterpret_cast<float4*>(dst + ((size_t)i * 256 + tid) * 16) = stg[i];
  };

  // DMA staging: the packed K/V images are a linear copy, so each stage is fetched by direct-to-LDS loads
  // (no VGPR staging, no ds_write pass); issued at the top of the previous stage, retired by the stage barrier.
  const unsigned lds0 = (unsigned)(size_t)lds;
  auto dma = [&](int stage, int buf) {
    if (tid < TS) stg_km = (stage * TS + tid < NT) ? gkmax[stage * TS + tid] : 0.f;
    const unsigned char* src = gkv + (size_t)stage * SG::STAGE_BYTES;
    const unsigned ldsb = lds0 + buf * SG::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < SG::CPT; ++i) {
      const size_t off = ((size_t)i * 256 + tid) * 16;
      if ((size_t)stage * SG::STAGE_BYTES + off < total_bytes && i * 256 + tid < SG::CHUNKS)
        __builtin_amdgcn_global_load_lds((const void*)(src + off), (lds_ptr_t)(size_t)(ldsb + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };
  if (DMA) {
    dma(0, 0);
    if (tid < TS) kmax_lds[tid] = stg_km;
  } else {
    gload(0);
    lstore(0);
  }
  __syncthreads();

  for (int st = 0; st < nstages; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstages) { if (DMA) dma(st + 1, buf ^ 1); else gload(st + 1); }
    const unsigned char* sb = lds + buf * SG::STAGE_BYTES;
    const int tiles_here = min(TS, NT - st * TS);
    // Stage-level bound: if |q| * max_k|k| over the WHOLE stage cannot exceed the running max by the threshold,
    // no tile of the stage can trigger a rescale -> run the stage without any per-tile max / vote / branch.
    float smax = 0.f;
#pragma unroll
    for (int t = 0; t < TS; ++t) smax = fmaxf(smax, kmax_lds[buf * TS + t]);
    bool fast = st > 0 && !(st == nstages - 1 && (N & 31) != 0);
#pragma unroll
    for (int u = 0; u < QS; ++u) fast = fast && !__any(fmaf(qn[u], smax, -m_run[u]) > RESCALE_THR);
    if (fast) {
      // Two tiles per iteration with ping-pong (A/B) fragment registers: the LDS reads of the next tile are issued
      // before the exp / pack / PV work of the current one, so the ds_read latency never sits in front of a tile.
      bf16x8 kA[G::KC], vA[G::DT][2], kB[G::KC], vB[G::DT][2];
      auto fetch = [&](int t, bf16x8 (&k_)[G::KC], bf16x8 (&v_)[G::DT][2]) {
        const unsigned char* tb = sb + t * G::TILE_BYTES;
#pragma unroll
        for (int c = 0; c < G::KC; ++c) k_[c] = *reinterpret_cast<const bf16x8*>(tb + c * 1024 + lane * 16);
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            v_[dt][s2] = *reinterpret_cast<const bf16x8*>(tb + G::KT_BYTES + ((dt * 2 + s2) * 2 + lh) * (G::RV * 16) + vrow * 16);
      };
      auto process = [&](const bf16x8 (&kf)[G::KC], const bf16x8 (&vf)[G::DT][2]) {
#pragma unroll
        for (int u = 0; u < QS; ++u) {
          f32x16 s = negm[u];
#pragma unroll
          for (int c = 0; c < G::KC; ++c) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[u][c], s, 0, 0, 0);
          bf16x8 pb[2];
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r]); pb[r >> 3][r & 7] = (__bf16)s[r]; }
          if (!G::ONES && DEN) {
            // denominators: v_dot2c_f32_bf16 with a (1, 1) operand adds TWO of the packed p per instruction -- 8 VALU ops per
            // tile instead of 16 scalar adds, and the sum is taken over the same bf16-rounded p the PV MFMA consumes
            sum_packed(pb, lacc[u]);
          } else if (!G::ONES) {
            osum[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pb[0], osum[u], 0, 0, 0);
            osum[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pb[1], osum[u], 0, 0, 0);
          }
#pragma unroll
          for (int dt = 0; dt < G::DT; ++dt) {
            o[u][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][0], pb[0], o[u][dt], 0, 0, 0);
            o[u][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][1], pb[1], o[u][dt], 0, 0, 0);
          }
        }
      };
      fetch(0, kA, vA);
      int t = 0;
      for (; t + 1 < tiles_here; t += 2) {
        fetch(t + 1, kB, vB);
        process(kA, vA);
        if (t + 2 < tiles_here) fetch(t + 2, kA, vA);
        process(kB, vB);
      }
      if (t < tiles_here) process(kA, vA);
    } else
    for (int t = 0; t < tiles_here; ++t) {
      const int kt = st * TS + t;
      const unsigned char* tb = sb + t * G::TILE_BYTES;
      const float kmx = kmax_lds[buf * TS + t];
      bf16x8 kf[G::KC];
#pragma unroll
      for (int c = 0; c < G::KC; ++c) kf[c] = *reinterpret_cast<const bf16x8*>(tb + c * 1024 + lane * 16);
      bf16x8 vf[G::DT][2];
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          vf[dt][s] = *reinterpret_cast<const bf16x8*>(tb + G::KT_BYTES + ((dt * 2 + s) * 2 + lh) * (G::RV * 16) + vrow * 16);

#pragma unroll
      for (int u = 0; u < QS; ++u) {
        f32x16 s = negm[u];
#pragma unroll
        for (int c = 0; c < G::KC; ++c) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[u][c], s, 0, 0, 0);
        const bool first = (kt == 0);
        const bool ragged = (kt == NT - 1) && (N & 31) != 0;
        // Cheap upper bound of every score of this tile: |q||k|max - m.  While it stays under the threshold no
        // key can force a rescale, so the exact row max (8 v_max3 + a cross-half exchange) is skipped.
        const float bound = fmaf(qn[u], kmx, -m_run[u]);
        if (first || ragged || __any(bound > RESCALE_THR)) {
          if (ragged) {   // padding keys of the last tile never win the max (and add nothing to the sums)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kt * 32 + acc_key(r, lh) >= N) s[r] = -INFINITY;
          }
          float tmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
          for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, s[r]), s[r + 1]);
          tmax = fmaxf(tmax, s[15]);
          tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
          if (first || __any(tmax > RESCALE_THR)) {
            // raise the running max of the rows that need it (every row on the first tile)
            const float delta = first ? tmax : (tmax > RESCALE_THR ? tmax : 0.f);
            m_run[u] += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] -= delta; negm[u][r] -= delta; }
            if (!first) {
              const float f = exp2f(-delta);
              lsum[u] *= f;
              lacc[u].x *= f; lacc[u].y *= f;
              if (!G::ONES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) osum[u][r] *= f;
              }
#pragma unroll
              for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[u][dt][r] *= f;
            }
          }
        }
        bf16x8 pb[2];
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __builtin_amdgcn_exp2f(s[r]);
          pb[r >> 3][r & 7] = (__bf16)p;
          ps += p;
        }
        if (!G::ONES && DEN) sum_packed(pb, lacc[u]);
        else if (!G::ONES) {
          osum[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pb[0], osum[u], 0, 0, 0);
          osum[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pb[1], osum[u], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) {
          o[u][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][0], pb[0], o[u][dt], 0, 0, 0);
          o[u][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][1], pb[1], o[u][dt], 0, 0, 0);
        }
      }
    }
    if (st + 1 < nstages) {
      // buffer buf^1 was last read in stage st-1, fenced by that stage's barrier
      if (DMA) { if (tid < TS) kmax_lds[(buf ^ 1) * TS + tid] = stg_km; }
      else lstore(buf ^ 1);
      __syncthreads();     // with LDS-DMA outstanding this also waits vmcnt(0): every wave's share has landed
    }
  }

  // ---- epilogue: O = O^T / l ; lane (q, half h) holds dd = (r&3) + 8(r>>2) + 4h + 32dt
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    float l;
    if (G::ONES) l = __shfl(o[u][0][(D / 8) * 4], lr);   // row D of O^T: register (D/8)*4 of lane half 0
    else if (DEN) { l = lacc[u].x + lacc[u].y; l += __shfl_xor(l, 32); }
    else l = osum[u][0];   // every row of the ones-product holds the full sum over keys of this lane's query
    const float inv = 1.0f / l;
    const int qr = q0 + 32 * u + lr;
    if (qr < N) {
      float* op = out + ((size_t)b * N + qr) * ldo + head * D;
      __bf16* op16 = out16 ? out16 + ((size_t)b * N + qr) * (size_t)(heads * D) + head * D : nullptr;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = dt * 32 + 8 * g + 4 * lh;
          if (dd < D) {
            const float4 v = make_float4(o[u][dt][4 * g] * inv, o[u][dt][4 * g + 1] * inv, o[u][dt][4 * g + 2] * inv,
                                         o[u][dt][4 * g + 3] * inv);
            if (op16) {
              typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
              bf16x4 w;
              w[0] = (__bf16)v.x; w[1] = (__bf16)v.y; w[2] = (__bf16)v.z; w[3] = (__bf16)v.w;
              *reinterpret_cast<bf16x4*>(op16 + dd) = w;
            } else {
              *reinterpret_cast<float4*>(op + dd) = v;
            }
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ flash, no running max
// Softmax needs the running max only to keep exp() in range.  When the scores of a whole (batch, head) pair are bounded
// -- |s| <= |q|max |k|max <= NM_BOUND in log2 units, known from the per-tile norms the packers emit -- p = 2^s itself is
// safe: p in [2^-80, 2^80], sums of 2^18 terms and the PV accumulators (times |v| up to 2^30) stay inside fp32's range, and bf16 / fp32
// relative precision does not depend on the scale.  This variant therefore runs with NO max: the S^T MFMA accumulates
// onto the constant 0 (no 16-register -m tile per query sub-tile), there is no bound test, no vote, no rescale path, and
// the kernel fits three waves per SIMD (the general kernel: 229 VGPRs, two).  The pair's score bound (attn_pair_bounds_k) routes
// each pair to exactly one of the two kernels; both are always launched and the workgroups of the other one return at once.

// bounds[bh].kmax = max |k| of a (batch, head) pair, blkq[bh][rb] = max |q|^2 of routing block rb (tpr 32-query tiles),
// bounds[bh].qmax2 = max |q|^2 of the pair -- fp32 bit patterns (non-negative floats order like unsigned integers, so the
// slices of a pair fold with atomicMax).  Norms of the bf16 operands as the MFMA sees them; q carries log2(e)/sqrt(d); the
// per-tile |k| bounds come from the packers.  Grid (slices, pairs); with more than one slice a memset node clears bounds[].
// Qmax2 != null: the producer of the operand images (the fused qkv projection) left the per-tile max |q|^2 there, so the
// 134 MB of Q are not read again and one workgroup per pair folds 2 x NT floats (no memset, no atomics).
template <int DP>
__global__ __launch_bounds__(256) void attn_block_bounds_k(const __bf16* __restrict__ Qp, const float* __restrict__ Kmax, int NT,
                                                           int tpr, int nrb, PairBounds* __restrict__ bounds,
                                                           float* __restrict__ blkq, const float* __restrict__ Qmax2) {
  __shared__ float rq[4], rk[4];
  const int bh = blockIdx.y, tid = threadIdx.x, nsl = gridDim.x, sl = blockIdx.x;
  const int rb0 = (int)((long long)nrb * sl / nsl), rb1 = (int)((long long)nrb * (sl + 1) / nsl);
  float pm = 0.f;                                    // max |q|^2 over this slice's blocks (for bounds[].qmax2)
  if (Qmax2) {
    for (int rb = rb0 + tid; rb < rb1; rb += 256) {
      float m = 0.f;
      const int t1 = min(NT, (rb + 1) * tpr);
      for (int t = rb * tpr; t < t1; ++t) { const float v = Qmax2[(size_t)bh * NT + t]; m = (v == v) ? fmaxf(m, v) : INFINITY; }
      blkq[(size_t)bh * nrb + rb] = m;
      pm = fmaxf(pm, m);
    }
  } else {
    const __bf16* q = Qp + (size_t)bh * NT * 32 * DP;
    for (int rb = rb0; rb < rb1; ++rb) {
      const int r1 = min(NT, (rb + 1) * tpr) * 32;
      float m = 0.f;
      for (int r = rb * tpr * 32 + tid; r < r1; r += 256) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < DP; c += 8) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(q + (size_t)r * DP + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; ss = fmaf(f, f, ss); }
        }
        m = (ss == ss) ? fmaxf(m, ss) : INFINITY;    // NaN operands: route to the general kernel
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      __syncthreads();
      if ((tid & 63) == 0) rq[tid >> 6] = m;
      __syncthreads();
      m = fmaxf(fmaxf(rq[0], rq[1]), fmaxf(rq[2], rq[3]));
      if (tid == 0) blkq[(size_t)bh * nrb + rb] = m;
      pm = fmaxf(pm, m);
    }
  }
  float km = 0.f;
  const int t0 = (int)((long long)NT * sl / nsl), t1 = (int)((long long)NT * (sl + 1) / nsl);
  for (int t = t0 + tid; t < t1; t += 256) { const float v = Kmax[(size_t)bh * NT + t]; km = (v == v) ? fmaxf(km, v) : INFINITY; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { pm = fmaxf(pm, __shfl_xor(pm, off)); km = fmaxf(km, __shfl_xor(km, off)); }
  __syncthreads();
  if ((tid & 63) == 0) { rq[tid >> 6] = pm; rk[tid >> 6] = km; }
  __syncthreads();
  if (tid == 0) {
    pm = fmaxf(fmaxf(rq[0], rq[1]), fmaxf(rq[2], rq[3]));
    km = fmaxf(fmaxf(rk[0], rk[1]), fmaxf(rk[2], rk[3]));
    if (nsl == 1) { bounds[bh].qmax2 = __float_as_uint(pm); bounds[bh].kmax = __float_as_uint(km); }
    else {
      atomicMax(&bounds[bh].qmax2, __float_as_uint(pm));
      atomicMax(&bounds[bh].kmax, __float_as_uint(km));
    }
  }
}

template <int D, int QS, int TS, int SDEN = 0>
__global__ __launch_bounds__(256, 3) void flash_bf16_nm(const __bf16* __restrict__ Qp, const unsigned char* __restrict__ KVp,
                                                        const PairBounds* __restrict__ bounds, float* __restrict__ out, int ldo,
                                                        int N, int NT, int heads, int nqb, int npairs, __bf16* __restrict__ out16,
                                                        const float* __restrict__ blkq, float nm_bound) {
  using G = AttnGeom<D>;
  using SG = StageGeom<D, TS>;
  // head_dim <= 16: the denominator comes out of the ones row of V^T.  head_dim 32 has no spare accumulator row: either an
  // all-ones A operand feeds a second accumulator tile whose every row is the column sum of P (SDEN = 0: two more MFMAs per
  // 32 keys, six in all -- matrix pipe 192 cycles per 1024 scores beside 212 of VALU issue) or the lane adds its sixteen p's
  // (SDEN = 1: 64 more VALU cycles, 260 in all, matrix pipe 128).  Measured at B = 16, N = 16 384, 4 heads (tools/attn_ab32.py,
  // profiles/r02_attn_nomax_headdim32_ab.txt): general kernel 2.77 ms, ones-MFMA 2.78, scalar adds 2.61 (QS = 2, 16 KiB
  // stages) -- 318 cycles per 1024 scores against the 199 of head_dim 16: with both pipes this full they overlap poorly.
  // N % 32 == 0 there (host check): without a ones row nothing masks the padding keys of a ragged last tile out of the sum.
  static_assert(G::DT <= 2 && G::KC <= 4, "no-max variant: head_dim <= 64");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 stage buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int wg = blockIdx.x, bh, qb;
  if ((npairs & 7) == 0) {
    int xcd = wg & 7, idx = wg >> 3;
    bh = (idx / nqb) * 8 + xcd;
    qb = idx - (idx / nqb) * nqb;
  } else {
    bh = wg / nqb;
    qb = wg - bh * nqb;
  }
  if (!block_is_bounded(bounds, blkq, bh, nqb, qb, nm_bound)) return;   // this query block goes through the general kernel
  const int b = bh / heads, head = bh - b * heads;
  const int q0 = qb * (128 * QS) + wave * (32 * QS);
  bf16x8 qf[QS][G::KC];
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    int qr = q0 + 32 * u + lr;
    if (qr >= NT * 32) qr = NT * 32 - 1;
#pragma unroll
    for (int c = 0; c < G::KC; ++c)
      qf[u][c] = *reinterpret_cast<const bf16x8*>(Qp + ((size_t)bh * NT * 32 + qr) * G::DP + 16 * c + 8 * lh);
  }
  constexpr bool LMFMA = !G::ONES && !SDEN;              // denominator: ones-operand MFMA tile, or (SDEN) scalar adds
  f32x16 o[QS][G::DT], lsum[LMFMA ? QS : 1];
  float lacc[QS][2];
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    lacc[u][0] = lacc[u][1] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) o[u][dt][r] = 0.f;
      if (LMFMA) lsum[u][r] = 0.f;
    }
  }
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  bf16x8 ones8;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones8[j] = (__bf16)1.0f;

  const unsigned char* gkv = KVp + (size_t)bh * NT * G::TILE_BYTES;
  const int nstages = (NT + TS - 1) / TS;
  const size_t total_bytes = (size_t)NT * G::TILE_BYTES;
  const int vrow = vt_row<D, G::ONES>(lr);                 // row of the V^T slab this lane reads (row D = ones)
  const unsigned lds0 = (unsigned)(size_t)lds;
  auto dma = [&](int stage, int buf) {
    const unsigned char* src = gkv + (size_t)stage * SG::STAGE_BYTES;
    const unsigned ldsb = lds0 + buf * SG::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < SG::CPT; ++i) {
      const size_t off = ((size_t)i * 256 + tid) * 16;
      if ((size_t)stage * SG::STAGE_BYTES + off < total_bytes && i * 256 + tid < SG::CHUNKS)
        __builtin_amdgcn_global_load_lds((const void*)(src + off), (lds_ptr_t)(size_t)(ldsb + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };
  dma(0, 0);
  __syncthreads();
  for (int st = 0; st < nstages; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstages) dma(st + 1, buf ^ 1);
    const unsigned char* sb = lds + buf * SG::STAGE_BYTES;
    const int tiles_here = min(TS, NT - st * TS);
    bf16x8 kA[G::KC], vA[G::DT][2], kB[G::KC], vB[G::DT][2];
    auto fetch = [&](int t, bf16x8 (&k_)[G::KC], bf16x8 (&v_)[G::DT][2]) {
      const unsigned char* tb = sb + t * G::TILE_BYTES;
#pragma unroll
      for (int c = 0; c < G::KC; ++c) k_[c] = *reinterpret_cast<const bf16x8*>(tb + c * 1024 + lane * 16);
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          v_[dt][s2] = *reinterpret_cast<const bf16x8*>(tb + G::KT_BYTES + ((dt * 2 + s2) * 2 + lh) * (G::RV * 16) + vrow * 16);
    };
    auto process = [&](const bf16x8 (&kf)[G::KC], const bf16x8 (&vf)[G::DT][2]) {
#pragma unroll
      for (int u = 0; u < QS; ++u) {
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[u][0], zero16, 0, 0, 0);
#pragma unroll
        for (int c = 1; c < G::KC; ++c) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[u][c], s, 0, 0, 0);
        bf16x8 pb[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[r]);
          if constexpr (!G::ONES && SDEN) lacc[u][r & 1] += pv;
          pb[r >> 3][r & 7] = (__bf16)pv;
        }
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) {
          o[u][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][0], pb[0], o[u][dt], 0, 0, 0);
          o[u][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][1], pb[1], o[u][dt], 0, 0, 0);
        }
        if constexpr (LMFMA) {
          lsum[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pb[0], lsum[u], 0, 0, 0);
          lsum[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pb[1], lsum[u], 0, 0, 0);
        }
      }
    };
    fetch(0, kA, vA);
    int t = 0;
    for (; t + 1 < tiles_here; t += 2) {
      fetch(t + 1, kB, vB);
      process(kA, vA);
      if (t + 2 < tiles_here) fetch(t + 2, kA, vA);
      process(kB, vB);
    }
    if (t < tiles_here) process(kA, vA);
    if (st + 1 < nstages) __syncthreads();     // with LDS-DMA outstanding this also waits vmcnt(0): every wave's share has landed
  }
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    // row D of O^T: register (D/8)*4 of lane half 0; or any row of the ones-operand tile (its column sums of P)
    float l;
    if constexpr (G::ONES) l = __shfl(o[u][0][(D / 8) * 4], lr);
    else if constexpr (LMFMA) l = lsum[LMFMA ? u : 0][0];
    else { l = lacc[u][0] + lacc[u][1]; l += __shfl_xor(l, 32); }   // the lane halves hold the two key halves of a query
    const float inv = 1.0f / l;
    const int qr = q0 + 32 * u + lr;
    if (qr < N) {
      float* op = out + ((size_t)b * N + qr) * ldo + head * D;
      __bf16* op16 = out16 ? out16 + ((size_t)b * N + qr) * (size_t)(heads * D) + head * D : nullptr;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * dt + 8 * g + 4 * lh;
        if (dd < D) {
          const float4 v = make_float4(o[u][dt][4 * g] * inv, o[u][dt][4 * g + 1] * inv, o[u][dt][4 * g + 2] * inv, o[u][dt][4 * g + 3] * inv);
          if (op16) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 w;
            w[0] = (__bf16)v.x; w[1] = (__bf16)v.y; w[2] = (__bf16)v.z; w[3] = (__bf16)v.w;
            *reinterpret_cast<bf16x4*>(op16 + dd) = w;
          } else {
            *reinterpret_cast<float4*>(op + dd) = v;
          }
        }
      }
    }
  }
}

template <int D, int QS, int TS = AttnGeom<D>::TS_DEFAULT, int WPS = 0, int DMA = 1, int DEN = 0, int NMTS = 0, int NMQS = 2, int NMSD = 0>
void run_flash(const float* qkv, float* out, int ldo, int B, int N, int C, int heads, void* ws, hipStream_t s,
               void* out16, const AttnOpts& opt) {
  using G = AttnGeom<D>;
  using SG = StageGeom<D, TS>;
  const int NT = (N + 31) / 32;
  const int BH = B * heads;
  __bf16* Qp = static_cast<__bf16*>(ws);
  size_t q_bytes = ((size_t)BH * NT * 32 * G::DP * 2 + 255) & ~size_t(255);
  unsigned char* KVp = static_cast<unsigned char*>(ws) + q_bytes;
  const size_t kv_bytes = ((size_t)BH * NT * G::TILE_BYTES + 255) & ~size_t(255);
  float* Kmax = reinterpret_cast<float*>(KVp + kv_bytes);
  const float qscale = 1.44269504088896340736f / sqrtf((float)D);
  if (qkv) hipLaunchKernelGGL(pack_qkv_bf16<D>, dim3(NT, heads, B), dim3(256), 0, s, qkv, Qp, KVp, Kmax, N, NT, C, qscale);
  const int nqb = (N + 128 * QS - 1) / (128 * QS);
  static PerDeviceOnce attr;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bf16<D, QS, TS, WPS, DMA, DEN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SG::STAGE_BYTES + 2 * TS * 4));
  });
  const PairBounds* skip = nullptr;
  const float* blkq_c = nullptr;
  int nrb = 1, rbq = 1;
  if constexpr (NMTS > 0) {
    // query blocks with bounded scores run without a running max (three waves per SIMD); the general kernel takes the rest
    using SN = StageGeom<D, NMTS>;
    static_assert((128 * NMQS) % (128 * QS) == 0, "a routing block (one no-max workgroup) must be whole general workgroups");
    PairBounds* flags = reinterpret_cast<PairBounds*>(Kmax + (size_t)BH * NT);
    float* Qmax2w = reinterpret_cast<float*>(flags + BH);                  // per-tile max |q|^2 (written by the fused projection)
    float* blkq = Qmax2w + (size_t)BH * NT;                                // per-routing-block max |q|^2
    const float* Qmax2 = qkv ? nullptr : Qmax2w;
    constexpr int TPR = 4 * NMQS;                                          // 32-query tiles per routing block
    const int nqb_nm = (N + 128 * NMQS - 1) / (128 * NMQS);
    nrb = nqb_nm; rbq = NMQS / QS;
    const int nsl = (!Qmax2 && NT >= 256) ? 32 : 1;
    if (nsl > 1) DDPMIR_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)BH * sizeof(PairBounds), s));
    hipLaunchKernelGGL(attn_block_bounds_k<G::DP>, dim3(nsl, BH), dim3(256), 0, s, (const __bf16*)Qp, (const float*)Kmax, NT, TPR, nrb, flags,
                       blkq, Qmax2);
    static PerDeviceOnce attr_nm;
    attr_nm.run([] {
      DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bf16_nm<D, NMQS, NMTS, NMSD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SN::STAGE_BYTES));
    });
    hipLaunchKernelGGL((flash_bf16_nm<D, NMQS, NMTS, NMSD>), dim3((unsigned)(nqb_nm * BH)), dim3(256), 2 * SN::STAGE_BYTES, s, (const __bf16*)Qp,
                       (const unsigned char*)KVp, (const PairBounds*)flags, out, ldo, N, NT, heads, nqb_nm, BH, reinterpret_cast<__bf16*>(out16),
                       (const float*)blkq, opt.nm_bound);
    skip = flags;
    blkq_c = blkq;
    if (opt.general_total) *opt.general_total += (int64_t)nqb * BH;
  }
  hipLaunchKernelGGL((flash_bf16<D, QS, TS, WPS, DMA, DEN>), dim3((unsigned)(nqb * BH)), dim3(256), 2 * SG::STAGE_BYTES + 2 * TS * 4, s, Qp, KVp, Kmax,
                     out, ldo, N, NT, heads, nqb, BH, reinterpret_cast<__bf16*>(out16), skip, blkq_c, nrb, rbq, opt.nm_bound, opt.general_taken);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

template <int D>
size_t ws_bytes_for(int B, int N, int heads) {
  using G = AttnGeom<D>;
  const size_t NT = (N + 31) / 32, BH = (size_t)B * heads;
  // ... + per-tile |k| bounds + the pairs' bounds + per-tile max |q|^2 + per-routing-block max |q|^2 (<= one per tile)
  return ((BH * NT * 32 * G::DP * 2 + 255) & ~size_t(255)) + ((BH * NT * G::TILE_BYTES + 255) & ~size_t(255)) +
         BH * NT * 4 + BH * 8 + BH * NT * 4 + BH * NT * 4 + 256;
}

}  // namespace

bool attention_bf16_supported(int N, int C, int heads) {
  const int D = C / heads;
  return (D == 8 || D == 16 || D == 32 || D == 64 || D == 128) && N >= 64;
}

static_assert(AttnGeom<16>::TILE_BYTES == ATTN16_TILE_BYTES && AttnGeom<16>::KT_BYTES == ATTN16_KT_BYTES &&
              AttnGeom<16>::RV == ATTN16_RV && AttnGeom<16>::DP == 16, "AttnPack16 geometry out of sync");

bool attention_bf16_prepack_supported(int N, int C, int heads) {
  const int D = heads > 0 ? C / heads : 0;
  return heads > 0 && heads * D == C && (D == 8 || D == 16 || D == 32) && (C % 64) == 0 && (N % 32) == 0 && N >= 64;
}

template <int D>
static AttnPack16 pack_targets_for(int B, int N, int heads, void* ws) {   // same carve-up as run_flash<D>
  using G = AttnGeom<D>;
  const size_t NT = (N + 31) / 32, BH = (size_t)B * heads;
  const size_t q_bytes = (BH * NT * 32 * G::DP * 2 + 255) & ~size_t(255);
  const size_t kv_bytes = (BH * NT * G::TILE_BYTES + 255) & ~size_t(255);
  AttnPack16 t;
  t.q = ws;
  t.kv = static_cast<unsigned char*>(ws) + q_bytes;
  t.kmax = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + q_bytes + kv_bytes);
  t.qmax2 = reinterpret_cast<float*>(reinterpret_cast<PairBounds*>(t.kmax + BH * NT) + BH);   // behind the pairs' bounds (run_flash)
  t.ntiles = (int)NT;
  t.qscale = 1.44269504088896340736f / sqrtf((float)D);
  t.D = D;
  t.heads = heads;
  return t;
}

static_assert(AttnGeom<32>::TILE_BYTES == 4096 && AttnGeom<32>::KT_BYTES == 2048 && AttnGeom<32>::RV == 32 && AttnGeom<32>::DP == 32,
              "head_dim-32 pack geometry out of sync with gemm16.hip");
static_assert(AttnGeom<8>::TILE_BYTES == 1024 + 4 * 9 * 16 && AttnGeom<8>::KT_BYTES == 1024 && AttnGeom<8>::RV == 9 && AttnGeom<8>::DP == 16,
              "head_dim-8 pack geometry out of sync with gemm16.hip");

AttnPack16 attention_bf16_pack_targets(int B, int N, int C, int heads, void* ws) {
  switch (C / heads) {
    case 8: return pack_targets_for<8>(B, N, heads, ws);
    case 32: return pack_targets_for<32>(B, N, heads, ws);
    default: return pack_targets_for<16>(B, N, heads, ws);
  }
}

size_t attention_bf16_workspace(int B, int N, int C, int heads) {
  switch (C / heads) {
    case 8: return ws_bytes_for<8>(B, N, heads);
    case 16: return ws_bytes_for<16>(B, N, heads);
    case 32: return ws_bytes_for<32>(B, N, heads);
    case 64: return ws_bytes_for<64>(B, N, heads);
    case 128: return ws_bytes_for<128>(B, N, heads);
    default: return 0;
  }
}

void launch_attention_bf16(const float* qkv, float* out, int ldo, int B, int N, int C, int heads, void* workspace,
                           hipStream_t s, void* out16, const AttnOpts* opts) {
  const AttnOpts opt = opts ? *opts : AttnOpts();
  if (!workspace || (!out && !out16)) throw std::logic_error("attention_bf16: workspace / output pointer missing");
#ifdef DDPMIR_EXPERIMENTS   // tile-shape A/B instantiations (tools/attn_ab*.py); the shipped library has one kernel per head_dim
  const int var = exp_value("DDPMIR_ATTN_VARIANT");
  if (C / heads == 32 && var >= 20) {   // head_dim 32 experiments
    switch (var) {
      case 20: run_flash<32, 1, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 21: run_flash<32, 1, 8>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 22: run_flash<32, 2, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 23: run_flash<32, 2, 8>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 24: run_flash<32, 1, 16>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 25: run_flash<32, 1, 8, 0, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 26: run_flash<32, 2, 8, 0, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 27: run_flash<32, 1, 8, 3, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 30: run_flash<32, 2, 4, 0, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 50: run_flash<32, 2, 8, 0, 1, 1, 4, 2, 0>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 51: run_flash<32, 2, 8, 0, 1, 1, 4, 2, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 52: run_flash<32, 2, 8, 0, 1, 1, 4, 4, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 54: run_flash<32, 2, 8, 0, 1, 1, 8, 2, 0>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 55: run_flash<32, 2, 8, 0, 1, 1, 2, 4, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 56: run_flash<32, 2, 8, 0, 1, 1, 2, 2, 0>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 31: run_flash<32, 2, 16, 0, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 28: run_flash<32, 1, 4, 0, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 29: run_flash<32, 1, 6, 0, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      default: break;
    }
  }
  if (C / heads == 16 && var != 0) {
    switch (var) {
      case 1: run_flash<16, 1, 16>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 2: run_flash<16, 1, 8>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 3: run_flash<16, 1, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 4: run_flash<16, 2, 16>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 5: run_flash<16, 2, 8>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 6: run_flash<16, 2, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 7: run_flash<16, 2, 16, 0, 0>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 8: run_flash<16, 2, 8, 0, 0>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 9: run_flash<16, 1, 16, 0, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 10: run_flash<16, 2, 8, 3, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 11: run_flash<16, 2, 4, 3, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 14: run_flash<16, 2, 16, 0, 1, 0, 16>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;   // no-max kernel, 2 WG/CU (LDS)
      case 15: run_flash<16, 2, 16, 0, 1, 0, 12>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;   // 3 WG/CU
      case 16: run_flash<16, 2, 16, 0, 1, 0, 8>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;    // 4 WG/CU
      case 17: run_flash<16, 2, 16, 0, 1, 0, 6>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;    // 4 WG/CU, shorter stages
      case 18: run_flash<16, 2, 16>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;                // the general kernel alone
      case 21: run_flash<16, 2, 16, 0, 1, 0, 8, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;   // 4 sub-tiles
      case 22: run_flash<16, 2, 16, 0, 1, 0, 10, 2>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 23: run_flash<16, 2, 16, 0, 1, 0, 4, 2>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 24: run_flash<16, 2, 16, 0, 1, 0, 12, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      case 25: run_flash<16, 2, 16, 0, 1, 0, 6, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); return;
      default: break;
    }
  }
#endif
  switch (C / heads) {
    // head_dim <= 16 is bound by v_exp issue: two query sub-tiles per wave share every K/V fragment read and give
    // the scheduler two independent exp/pack/PV chains (A/B on one device: 7.92 vs 8.49 ms at N = 65 536)
    // ... and 16-tile (512-key) stages: one barrier per 32 K scores per wave (A/B: TS 4/8/16 = 8.86/7.66/7.24 ms)
    // ... bounded scores (the normal case) take the no-max kernel: four query sub-tiles per wave at three waves per SIMD
    // (A/B on one device, B=4, N=65 536: general kernel 7.30 ms; no-max QS/TS 2/12 6.96, 2/8 6.97, 3/8 7.09, 4/8 6.84)
    case 8: run_flash<8, 2, 16, 0, 1, 0, 8, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); break;
    case 16: run_flash<16, 2, 16, 0, 1, 0, 8, 4>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); break;
    case 32:   // bounded scores: no-max kernel at 3 waves/SIMD (16 KiB stages, scalar-add denominators); general kernel otherwise
      if ((N % 32) == 0) run_flash<32, 2, 8, 0, 1, 1, 4, 2, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt);
      else run_flash<32, 2, 8, 0, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt);
      break;
    case 64:   // bounded scores: no-max kernel (one query sub-tile per wave, 16 KiB stages, scalar-add denominators): 355 -> 269 us
      // at N = 4096, B = 16; below N = 2048 its bounds pass and memset cost more than it saves
      if ((N % 32) == 0 && N >= 2048) run_flash<64, 1, AttnGeom<64>::TS_DEFAULT, 0, 1, 0, 2, 1, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt);
      else run_flash<64, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt);
      break;
    case 128: run_flash<128, 1>(qkv, out, ldo, B, N, C, heads, workspace, s, out16, opt); break;
    default: throw std::runtime_error("attention_bf16: unsupported head dim");
  }
}

}  // namespace ddpmir

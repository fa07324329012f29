// Self-attention core softmax(Q K^T / sqrt(d)) V for the per-block nn.MultiheadAttention
// (reference: webp_inference.py:295,317-321; every one of the 13 blocks attends over all H*W tokens).
//
// flash_f32<D>: online-softmax tiling, exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32), never materialises
// the N x N scores (64 GiB/image at 256^2).  One wave owns 32 query rows; everything is keyed by the query
// on the LANE (lane & 31):
//   S^T[key][q] = K Q^T      A = K tile from LDS (lane: key row, 4 consecutive d), B = Q (registers)
//   the 32x32 accumulator gives each lane 16 keys of ITS query  -> row max / row sum are in-lane (+1 xor-32)
//   O^T[dd][q] += V^T P^T    A = V tile from LDS (lane: dd column), B = the P registers as they stand:
//   accumulator register s of lane half h holds key (s&3)+8(s>>2)+4h, which is exactly the k index the
//   32x32x2 MFMA takes from lane half h at step s -- no shuffle, no LDS round trip for P.
// small_f32: one wave per query row, scores in LDS; used for tiny maps (N < 64) and head_dim > 128
// (bottleneck blocks, N <= 1024).
#include <cstdlib>

#include "kernels.h"

namespace ddpmir {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int KT = 32;   // keys per tile

template <int D>
__global__ __launch_bounds__(256) void flash_f32(const float* __restrict__ qkv, float* __restrict__ out, int ldo,
                                                 int N, int C, float qscale) {
  constexpr int DP = D >= 256 ? D : D + 4;   // padded K row (conflict-free b128 reads); 256: 64 KiB of static LDS is the limit
  constexpr int DV = D < 32 ? 32 : D;        // V row width (>= one 32-wide dd tile)
  constexpr int DT = DV / 32;
  constexpr int QC = D / 8;                  // 8-wide d chunks
  __shared__ __attribute__((aligned(16))) float Ks[KT * DP];
  __shared__ __attribute__((aligned(16))) float Vs[KT * DV];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * C;
  const float* base = qkv + (size_t)b * N * ld + head * D;
  const int q_row = blockIdx.x * 128 + wave * 32 + lr;
  const bool q_ok = q_row < N;

  // Q fragment: comps 8c + 4h + j, pre-scaled by log2(e)/sqrt(d)
  float4 qf[QC];
  {
    const float* qp = base + (size_t)(q_ok ? q_row : 0) * ld + 4 * lh;
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      float4 v = *reinterpret_cast<const float4*>(qp + 8 * c);
      qf[c] = make_float4(v.x * qscale, v.y * qscale, v.z * qscale, v.w * qscale);
    }
  }
  if (D < 32) {  // zero the dd padding columns of V once
    for (int i = tid; i < KT * DV; i += 256) Vs[i] = 0.f;
  }

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (N + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();   // previous tile fully consumed (also orders the V zero fill)
    for (int idx = tid; idx < KT * (D / 4); idx += 256) {
      int row = idx / (D / 4), c4 = idx - row * (D / 4);
      int key = kt * KT + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < N) {
        const float* rp = base + (size_t)key * ld + 4 * c4;
        kv = *reinterpret_cast<const float4*>(rp + C);
        vv = *reinterpret_cast<const float4*>(rp + 2 * C);
      }
      *reinterpret_cast<float4*>(&Ks[row * DP + 4 * c4]) = kv;
      *reinterpret_cast<float4*>(&Vs[row * DV + 4 * c4]) = vv;
    }
    __syncthreads();

    // S^T tile: rows = keys, cols = queries
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      float4 ka = *reinterpret_cast<const float4*>(&Ks[lr * DP + 8 * c + 4 * lh]);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qf[c].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qf[c].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qf[c].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qf[c].w, s, 0, 0, 0);
    }
    // mask keys beyond N (only the last tile can have them)
    const int kbase = kt * KT + 4 * lh;
    if (kt == ntiles - 1 && (N & (KT - 1)) != 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kbase + (r & 3) + 8 * (r >> 2) >= N) s[r] = -INFINITY;
    }
    float tmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = exp2f(s[r] - m_new);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;   // per lane-half partial; halves are added at the end
    m_run = m_new;

#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int key = (st & 3) + 8 * (st >> 2) + 4 * lh;
        float va = Vs[key * DV + t * 32 + lr];
        o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, s[st], o[t], 0, 0, 0);
      }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    float* op = out + ((size_t)b * N + q_row) * ldo + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int dd = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (dd < D) op[dd] = o[t][r] * inv;
      }
  }
}

// head_dim 16 (the full-resolution levels, 95 % of the exact-fp32 FLOPs): v_mfma_f32_16x16x4_f32 tiles.  The 32x32x2
// tile above spends a 32-row PV product on 16 useful rows; here S^T (16 keys x 16 queries) and O^T (16 dims x 16 queries)
// are both exact fits: 8 MFMAs of 32 cycles per 256 scores instead of 12 x 64 per 1024 -> 1024 vs 1536 cycles per 1024.
// Layouts (lane l: g = l / 16, n = l % 16):  A[m = n][k = g], B[k = g][n], C/D[m = 4g + r][n], r = 0..3.
//   S^T = K Q^T:  A = K[key n][dim 4j + g] (LDS, dims permuted so the four j are one float4), B = Q[query n][dim 4j + g]
//   lane (g, n) then owns keys 4g..4g+3 of query n; used as B of the PV product step j it supplies key 4g + j, so the
//   A operand of step j is V[key 4g + j][dim n] (LDS holds V as [key group g][dim][j]: one float4 per lane and tile).
// One wave = 32 queries (two 16-query halves sharing every K/V fragment), exact online softmax per 16-key tile.
// head_dim 8 (the AVIF family's level 0, 8 heads) uses the same tiles with two k-steps and half of the O^T rows.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>   // D = 8 (AVIF level 0: half of the 16 output rows stay empty) or 16 (exact fit)
__global__ __launch_bounds__(256) void flash_f32_t16(const float* __restrict__ qkv, float* __restrict__ out, int ldo, int N,
                                                     int C, float qscale) {
  constexpr int KS = 64;                               // keys per LDS stage (4 tiles of 16)
  constexpr int NJ = D / 4;                            // k-steps of the S^T product
  __shared__ __attribute__((aligned(16))) float Ks[KS * D];       // [key][g][j]  = K[key][4j + g]
  __shared__ __attribute__((aligned(16))) float Vs[KS * 16];      // [key / 4][dim 0..15][key % 4] (dims >= D stay zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * C;
  const float* base = qkv + (size_t)b * N * ld + head * D;
  const int q0 = blockIdx.x * 128 + wave * 32;

  float qf[2][NJ];                                     // B operand of S^T: Q[query 16h + n][dim 4j + g], pre-scaled
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int qr = min(q0 + 16 * h + n, N - 1);
#pragma unroll
    for (int j = 0; j < NJ; ++j) qf[h][j] = base[(size_t)qr * ld + 4 * j + g] * qscale;
  }
  if (D < 16)
    for (int i = tid; i < KS * 16; i += 256) Vs[i] = 0.f;
  f32x4 o[2];
  float m_run[2], l_run[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { o[h] = f32x4{0.f, 0.f, 0.f, 0.f}; m_run[h] = -INFINITY; l_run[h] = 0.f; }

  const int nstages = (N + KS - 1) / KS;
  for (int st = 0; st < nstages; ++st) {
    __syncthreads();                                   // previous stage fully consumed (and the zero fill ordered)
    for (int idx = tid; idx < KS * NJ; idx += 256) {   // one float4 of K and of V per (key, 4-dim chunk)
      const int row = idx / NJ, c4 = idx - row * NJ, key = st * KS + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < N) {
        const float* rp = base + (size_t)key * ld + 4 * c4;
        kv = *reinterpret_cast<const float4*>(rp + C);
        vv = *reinterpret_cast<const float4*>(rp + 2 * C);
      }
      // K[key][4 c4 + e] -> Ks[key][g = e][j = c4];  V[key][4 c4 + e] -> Vs[key / 4][dim = 4 c4 + e][key % 4]
      Ks[row * D + 0 * NJ + c4] = kv.x; Ks[row * D + 1 * NJ + c4] = kv.y; Ks[row * D + 2 * NJ + c4] = kv.z; Ks[row * D + 3 * NJ + c4] = kv.w;
      float* vp = Vs + (row >> 2) * (16 * 4) + (row & 3);
      vp[(4 * c4 + 0) * 4] = vv.x; vp[(4 * c4 + 1) * 4] = vv.y; vp[(4 * c4 + 2) * 4] = vv.z; vp[(4 * c4 + 3) * 4] = vv.w;
    }
    __syncthreads();
    const int tiles = min(KS / 16, (N - st * KS + 15) / 16);
    for (int t = 0; t < tiles; ++t) {
      float ka[NJ];                                                                                // K[key n][4j + g], j < NJ
#pragma unroll
      for (int j = 0; j < NJ; ++j) ka[j] = Ks[(t * 16 + n) * D + g * NJ + j];
      const float4 va = *reinterpret_cast<const float4*>(&Vs[((t * 4 + g) * 16 + n) * 4]);        // V[key 4g + j][dim n]
      const int kbase = st * KS + t * 16 + 4 * g;                                                  // this lane's first key
      const bool ragged = kbase + 3 >= N;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j], qf[h][j], s, 0, 0, 0);
        if (ragged) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kbase + r >= N) s[r] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run[h], tmax);
        if (__any(m_new > m_run[h])) {                 // wave-uniform: rescale only when some row's maximum moved
          const float alpha = __builtin_amdgcn_exp2f(m_run[h] - m_new);   // raw v_exp_f32; exp2(-inf) = 0 on the first tile
          o[h] *= alpha;
          l_run[h] *= alpha;
          m_run[h] = m_new;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - m_run[h]);   // arguments <= 0: no range fix-up needed
        l_run[h] += (s[0] + s[1]) + (s[2] + s[3]);
        o[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, s[0], o[h], 0, 0, 0);
        o[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, s[1], o[h], 0, 0, 0);
        o[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, s[2], o[h], 0, 0, 0);
        o[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, s[3], o[h], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float l = l_run[h];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int qr = q0 + 16 * h + n;
    if (qr < N && 4 * g < D) {                         // lane (g, n) holds dims 4g..4g+3 of query n
      const float inv = 1.0f / l;
      *reinterpret_cast<float4*>(out + ((size_t)b * N + qr) * ldo + head * D + 4 * g) =
          make_float4(o[h][0] * inv, o[h][1] * inv, o[h][2] * inv, o[h][3] * inv);
    }
  }
}

// one wave per query row; scores of the row live in LDS (N <= SMALL_MAXN)
constexpr int SMALL_MAXN = 1024;

__global__ __launch_bounds__(256) void small_f32(const float* __restrict__ qkv, float* __restrict__ out, int ldo,
                                                 int N, int C, int D, float scale) {
  __shared__ float P[4][SMALL_MAXN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q_row = blockIdx.x * 4 + wave;
  if (q_row >= N) return;
  const int ld = 3 * C;
  const float* base = qkv + (size_t)b * N * ld + head * D;
  const float* qp = base + (size_t)q_row * ld;
  float mx = -INFINITY;
  for (int k = lane; k < N; k += 64) {
    const float* kp = base + (size_t)k * ld + C;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = fmaf(qp[d] * scale, kp[d], acc);
    P[wave][k] = acc;
    mx = fmaxf(mx, acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int k = lane; k < N; k += 64) {
    float e = __expf(P[wave][k] - mx);
    P[wave][k] = e;
    sum += e;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float* op = out + ((size_t)b * N + q_row) * ldo + head * D;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int k = 0; k < N; ++k) acc = fmaf(P[wave][k], base[(size_t)k * ld + 2 * C + d], acc);
    op[d] = acc * inv;
  }
}

}  // namespace

void launch_attention(const float* qkv, float* out, int ldo, int B, int N, int C, int heads, hipStream_t stream) {
  const int D = C / heads;
  if (D * heads != C || (D % 8) != 0) throw std::runtime_error("attention: head_dim must be a multiple of 8");
  const float scale = 1.0f / sqrtf((float)D);
  const bool flash_ok = (D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256) && N >= 64;
  if (!flash_ok) {
    if (N > SMALL_MAXN) throw std::runtime_error("attention: head_dim > 128 needs N <= 1024 tokens");
    dim3 grid(cdiv(N, 4), heads, B);
    hipLaunchKernelGGL(small_f32, grid, dim3(256), 0, stream, qkv, out, ldo, N, C, D, scale);
    DDPMIR_HIP_CHECK(hipGetLastError());
    return;
  }
  const float qscale = scale * 1.44269504088896340736f;
  dim3 grid(cdiv(N, 128), heads, B), block(256);
  switch (D) {
    case 8:
      if (exp_switch("DDPMIR_ATTN_F32_OLD")) { hipLaunchKernelGGL(flash_f32<8>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale); break; }
      hipLaunchKernelGGL(flash_f32_t16<8>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale);
      break;
    case 16:
      if (exp_switch("DDPMIR_ATTN_F32_OLD")) { hipLaunchKernelGGL(flash_f32<16>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale); break; }
      hipLaunchKernelGGL(flash_f32_t16<16>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale);
      break;
    case 32: hipLaunchKernelGGL(flash_f32<32>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale); break;
    case 64: hipLaunchKernelGGL(flash_f32<64>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale); break;
    case 128: hipLaunchKernelGGL(flash_f32<128>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale); break;
    // bottleneck blocks (C = 1024, 4 heads): 64..1024 tokens.  ~300 VGPRs (one wave per SIMD) but two orders of magnitude
    // fewer instructions than the scalar small_f32 path it replaces (1.1 ms -> see DESIGN 4 at 256 tokens)
    case 256: hipLaunchKernelGGL(flash_f32<256>, grid, block, 0, stream, qkv, out, ldo, N, C, qscale); break;
  }
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

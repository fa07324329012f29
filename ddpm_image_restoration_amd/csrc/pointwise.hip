// HBM-bound kernels of the U-Net forward: GroupNorm statistics, pooling, bilinear resampling, blockwise
// DCT / learned transforms, time embedding.  All NHWC fp32, 16-byte vector accesses along channels.
#include "kernels.h"

namespace ddpmir {

namespace {

// ------------------------------------------------------------------------------------------- GroupNorm
// Pass 1: a workgroup streams a contiguous run of pixels of ONE image, all channels (fully coalesced),
// every thread owns a fixed 4-channel slot -> one group; double partial sums per (b, split, g).
// Pass 2: fold the splits, emit scale = rstd*gamma, shift = beta - mean*scale per (b, c).
constexpr int GN_PIX_PER_SPLIT = 256;
constexpr int GN_MAX_SPLITS = 512;

__global__ __launch_bounds__(256) void gn_partial(const float* __restrict__ x, int ldx, int HW, int C, int G,
                                                  int splits, double* __restrict__ partial) {
  __shared__ double s1[256], s2[256];
  __shared__ int sg[256];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, sp = blockIdx.x;
  const int cpg = C / G;
  const int vec = (C % 4 == 0 && cpg % 4 == 0) ? 4 : 1;
  const int lanes_per_pix = C / vec;
  const int pix_per_iter = 256 / lanes_per_pix > 0 ? 256 / lanes_per_pix : 1;
  const int64_t p_begin = (int64_t)sp * HW / splits, p_end = (int64_t)(sp + 1) * HW / splits;
  double a1 = 0.0, a2 = 0.0;
  int g = -1;
  if (lanes_per_pix <= 256) {
    const int slot = tid % lanes_per_pix, pslot = tid / lanes_per_pix;
    if (pslot < pix_per_iter) {
      g = (slot * vec) / cpg;
      const float* xb = x + (size_t)b * HW * ldx + slot * vec;
      int64_t p = p_begin + pslot;
      if (vec == 4) {
        // 4 independent loads in flight per thread; fp32 partials of 4 values, folded into fp64 per pixel
        double b1 = 0.0, b2 = 0.0, c1 = 0.0, c2 = 0.0, d1 = 0.0, d2 = 0.0;
        for (; p + 3 * (int64_t)pix_per_iter < p_end; p += 4 * (int64_t)pix_per_iter) {
          float4 v0 = *reinterpret_cast<const float4*>(xb + p * ldx);
          float4 v1 = *reinterpret_cast<const float4*>(xb + (p + pix_per_iter) * ldx);
          float4 v2 = *reinterpret_cast<const float4*>(xb + (p + 2 * (int64_t)pix_per_iter) * ldx);
          float4 v3 = *reinterpret_cast<const float4*>(xb + (p + 3 * (int64_t)pix_per_iter) * ldx);
          a1 += (double)((v0.x + v0.y) + (v0.z + v0.w)); a2 += (double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w;
          b1 += (double)((v1.x + v1.y) + (v1.z + v1.w)); b2 += (double)v1.x * v1.x + (double)v1.y * v1.y + (double)v1.z * v1.z + (double)v1.w * v1.w;
          c1 += (double)((v2.x + v2.y) + (v2.z + v2.w)); c2 += (double)v2.x * v2.x + (double)v2.y * v2.y + (double)v2.z * v2.z + (double)v2.w * v2.w;
          d1 += (double)((v3.x + v3.y) + (v3.z + v3.w)); d2 += (double)v3.x * v3.x + (double)v3.y * v3.y + (double)v3.z * v3.z + (double)v3.w * v3.w;
        }
        a1 += b1 + c1 + d1; a2 += b2 + c2 + d2;
      }
      for (; p < p_end; p += pix_per_iter) {
        if (vec == 4) {
          float4 v = *reinterpret_cast<const float4*>(xb + p * ldx);
          a1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
          a2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
          float v = xb[p * ldx];
          a1 += v;
          a2 += (double)v * v;
        }
      }
    }
  } else {  // more 4-channel slots than threads (C > 1024): loop over slots
    // not needed for the supported models (C <= 1024)
  }
  s1[tid] = a1; s2[tid] = a2; sg[tid] = g;
  __syncthreads();
  if (tid < G) {
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < 256; ++i)
      if (sg[i] == tid) { t1 += s1[i]; t2 += s2[i]; }
    double* o = partial + (((size_t)b * splits + sp) * G + tid) * 2;
    o[0] = t1; o[1] = t2;
  }
}

// Wide groups (>= 32 channels = 128 contiguous bytes per pixel, levels 2-5): one workgroup per (sample, split, GROUP),
// so the deep levels with 64..1024 pixels still spread over G times more workgroups and the fold is a plain block
// reduction.  Same partial layout as gn_partial.
__global__ __launch_bounds__(256) void gn_partial_group(const float* __restrict__ x, int ldx, int HW, int C, int G,
                                                        int splits, double* __restrict__ partial) {
  __shared__ double r1[4], r2[4];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G, sp = blockIdx.x / G, b = blockIdx.y;
  const int cpg = C / G, lpp = cpg / 4;                 // lanes per pixel (8..256)
  const int ppi = 256 / lpp;                            // pixels per iteration
  const int slot = tid % lpp, pslot = tid / lpp;
  const int64_t p_begin = (int64_t)sp * HW / splits, p_end = (int64_t)(sp + 1) * HW / splits;
  double a1 = 0.0, a2 = 0.0;
  if (pslot < ppi) {
    const float* xb = x + (size_t)b * HW * ldx + g * cpg + slot * 4;
    for (int64_t p = p_begin + pslot; p < p_end; p += ppi) {
      const float4 v = *reinterpret_cast<const float4*>(xb + p * ldx);
      a1 += (double)((v.x + v.y) + (v.z + v.w));
      a2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); }
  if ((tid & 63) == 0) { r1[tid >> 6] = a1; r2[tid >> 6] = a2; }
  __syncthreads();
  if (tid == 0) {
    double* o = partial + (((size_t)b * splits + sp) * G + g) * 2;
    o[0] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    o[1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
  }
}

// one workgroup per (sample, group): strided sum of the double partials in a FIXED order (thread t takes slots t, t + 256, ...;
// wave shuffle tree; the four wave sums in order), then the group's channels get their folded scale / shift.  The order does
// not depend on how the partials were produced, so GroupNorm statistics are bit-reproducible run to run.
// n_per_group = elements behind one group's sums; scale / shift rows are ldsc floats apart; channels [C, zero_to) are zeroed.
__global__ __launch_bounds__(256) void gn_finalize(const double* __restrict__ partial, int splits, double n_per_group, int C, int ldsc,
                                                   int G, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ scale, float* __restrict__ shift, int zero_to) {
  __shared__ double r1[4], r2[4];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  double t1 = 0.0, t2 = 0.0;
  for (int s = tid; s < splits; s += 256) {
    const double* o = partial + (((size_t)b * splits + s) * G + g) * 2;
    t1 += o[0]; t2 += o[1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { t1 += __shfl_xor(t1, off); t2 += __shfl_xor(t2, off); }
  if ((tid & 63) == 0) { r1[tid >> 6] = t1; r2[tid >> 6] = t2; }
  __syncthreads();
  t1 = (r1[0] + r1[1]) + (r1[2] + r1[3]);
  t2 = (r2[0] + r2[1]) + (r2[2] + r2[3]);
  const double mean = t1 / n_per_group;
  double var = t2 / n_per_group - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + 1e-5));
  for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
    const float sc = rstd * gamma[c];
    scale[(size_t)b * ldsc + c] = sc;
    shift[(size_t)b * ldsc + c] = beta[c] - (float)mean * sc;
  }
  if (g == G - 1)
    for (int c = C + tid; c < zero_to; c += 256) { scale[(size_t)b * ldsc + c] = 0.f; shift[(size_t)b * ldsc + c] = 0.f; }
}

// ------------------------------------------------------------------------------------------- resampling
__global__ void maxpool2_k(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int H,
                           int W, int C4) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c4 = i % C4;
    int64_t p = i / C4;
    int xo = p % Wo; p /= Wo;
    int yo = p % Ho; int b = p / Ho;
    const float* ip = in + (((size_t)b * H + 2 * yo) * W + 2 * xo) * ldi + 4 * c4;
    float4 a = *reinterpret_cast<const float4*>(ip);
    float4 bq = *reinterpret_cast<const float4*>(ip + ldi);
    float4 c = *reinterpret_cast<const float4*>(ip + (size_t)W * ldi);
    float4 d = *reinterpret_cast<const float4*>(ip + (size_t)W * ldi + ldi);
    float4 r = make_float4(fmaxf(fmaxf(a.x, bq.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, bq.y), fmaxf(c.y, d.y)),
                           fmaxf(fmaxf(a.z, bq.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, bq.w), fmaxf(c.w, d.w)));
    *reinterpret_cast<float4*>(out + (((size_t)b * Ho + yo) * Wo + xo) * ldo + 4 * c4) = r;
  }
}

// torch upsample_bilinear2d, align_corners=False: src = max(scale*(dst+0.5)-0.5, 0)
__device__ __forceinline__ void lin_coord(int dst, int in_size, float scale, int& i0, int& i1, float& w0, float& w1) {
  float src = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  i0 = min((int)src, in_size - 1);
  i1 = min(i0 + 1, in_size - 1);
  w1 = src - (float)i0;
  w0 = 1.0f - w1;
}

__global__ void bilinear_k(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int Hi,
                           int Wi, int Ho, int Wo, int C4, float sh, float sw, int accumulate, float out_scale) {
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c4 = i % C4;
    int64_t p = i / C4;
    int xo = p % Wo; p /= Wo;
    int yo = p % Ho; int b = p / Ho;
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    lin_coord(yo, Hi, sh, y0, y1, wy0, wy1);
    lin_coord(xo, Wi, sw, x0, x1, wx0, wx1);
    const float* ib = in + (size_t)b * Hi * Wi * ldi + 4 * c4;
    float4 v00 = *reinterpret_cast<const float4*>(ib + ((size_t)y0 * Wi + x0) * ldi);
    float4 v01 = *reinterpret_cast<const float4*>(ib + ((size_t)y0 * Wi + x1) * ldi);
    float4 v10 = *reinterpret_cast<const float4*>(ib + ((size_t)y1 * Wi + x0) * ldi);
    float4 v11 = *reinterpret_cast<const float4*>(ib + ((size_t)y1 * Wi + x1) * ldi);
    float4 r;
    r.x = wy0 * (wx0 * v00.x + wx1 * v01.x) + wy1 * (wx0 * v10.x + wx1 * v11.x);
    r.y = wy0 * (wx0 * v00.y + wx1 * v01.y) + wy1 * (wx0 * v10.y + wx1 * v11.y);
    r.z = wy0 * (wx0 * v00.z + wx1 * v01.z) + wy1 * (wx0 * v10.z + wx1 * v11.z);
    r.w = wy0 * (wx0 * v00.w + wx1 * v01.w) + wy1 * (wx0 * v10.w + wx1 * v11.w);
    float* op = out + (((size_t)b * Ho + yo) * Wo + xo) * ldo + 4 * c4;
    if (accumulate) {
      float4 o = *reinterpret_cast<float4*>(op);
      r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    r.x *= out_scale; r.y *= out_scale; r.z *= out_scale; r.w *= out_scale;
    *reinterpret_cast<float4*>(op) = r;
  }
}

// ------------------------------------------------------------------------------------------- pool / upsample with statistics
// The tensors BETWEEN residual blocks (bf16 / fp8 mode): 2x2 max-pool, bilinear x2, or statistics alone over NHWC with 16-byte
// accesses, for an fp32 stream (T = float, 4 channels per lane) or the bf16 activation stream (T = __bf16, 8 channels per lane;
// results rounded to bf16 as the reference's autocast rounds every conv output):
//   STATS  no output, sums of `in`          POOL  out = maxpool2(in)          UP  out = bilinear x2 (in)
// When `partial` is given the pass also leaves the GroupNorm partial sums of the tensor it produces (of the values a consumer
// will read), in gn_partial's layout [b][splits][G][2]: the statistics pass over a pooled / upsampled block input disappears.
// A workgroup owns a run of output pixels of one sample; thread = (pixel slot, channel slot), so a thread's channels lie in
// one group; lanes fold by xor-shuffles over the offsets that stay inside a group or step whole pixels.  Fixed order: the
// statistics are bit-reproducible.
typedef __bf16 h8_t __attribute__((ext_vector_type(8)));
enum { S16_STATS = 0, S16_POOL = 1, S16_UP = 2 };

template <typename T> struct LaneVec;
template <> struct LaneVec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float v[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, float v[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct LaneVec<__bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const __bf16* p, float v[8]) {
    const h8_t t = *reinterpret_cast<const h8_t*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
  }
  static __device__ __forceinline__ void store(__bf16* p, float v[8]) {   // v becomes what was stored
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = (__bf16)v[j]; v[j] = (float)o[j]; }
    *reinterpret_cast<h8_t*>(p) = o;
  }
};

template <int MODE, typename T>
__global__ __launch_bounds__(256) void stream_k(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo, int Hi, int Wi,
                                                int CL /* lanes per pixel */, int gl /* lanes per group */, int G, int splits, FastDiv fd_wo,
                                                double* __restrict__ partial) {
  constexpr int N = LaneVec<T>::N;
  __shared__ double red[4][8][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  // XCD-aware placement (workgroup id mod 8 = XCD): each XCD takes a contiguous band of pixel runs, so the two output rows that
  // share a pair of input rows of the bilinear pass hit the same L2 (round-robin runs re-fetched the taps from HBM: PMC 0.97 GB
  // read for 0.27 GB of input at 512 x 512, B = 8)
  int sp = blockIdx.x;
  if ((splits & 7) == 0) sp = (sp & 7) * (splits >> 3) + (sp >> 3);
  const int Ho = MODE == S16_POOL ? Hi / 2 : MODE == S16_UP ? 2 * Hi : Hi;
  const int Wo = MODE == S16_POOL ? Wi / 2 : MODE == S16_UP ? 2 * Wi : Wi;
  const int HWo = Ho * Wo;
  const int ppi = 256 / CL, slot = tid % CL, pslot = tid / CL;
  const int p_begin = (int)((int64_t)sp * HWo / splits), p_end = (int)((int64_t)(sp + 1) * HWo / splits);
  const T* ib = in + (size_t)b * Hi * Wi * ldi + slot * N;
  double a1 = 0.0, a2 = 0.0;
  for (int p = p_begin + pslot; p < p_end; p += ppi) {
    float v[N];
    if (MODE == S16_STATS) {
      LaneVec<T>::load(ib + (size_t)p * ldi, v);
    } else {
      const int yo = (int)fd_wo.div((unsigned)p), xo = p - yo * Wo;
      if (MODE == S16_POOL) {
        const T* q = ib + ((size_t)(2 * yo) * Wi + 2 * xo) * ldi;
        float t0[N], t1[N], t2[N];
        LaneVec<T>::load(q, v); LaneVec<T>::load(q + ldi, t0);
        LaneVec<T>::load(q + (size_t)Wi * ldi, t1); LaneVec<T>::load(q + (size_t)Wi * ldi + ldi, t2);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = fmaxf(fmaxf(v[j], t0[j]), fmaxf(t1[j], t2[j]));
      } else {
        int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
        lin_coord(yo, Hi, 0.5f, y0, y1, wy0, wy1);
        lin_coord(xo, Wi, 0.5f, x0, x1, wx0, wx1);
        float v00[N], v01[N], v10[N], v11[N];
        LaneVec<T>::load(ib + ((size_t)y0 * Wi + x0) * ldi, v00); LaneVec<T>::load(ib + ((size_t)y0 * Wi + x1) * ldi, v01);
        LaneVec<T>::load(ib + ((size_t)y1 * Wi + x0) * ldi, v10); LaneVec<T>::load(ib + ((size_t)y1 * Wi + x1) * ldi, v11);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = wy0 * (wx0 * v00[j] + wx1 * v01[j]) + wy1 * (wx0 * v10[j] + wx1 * v11[j]);
      }
      LaneVec<T>::store(out + ((size_t)b * HWo + p) * ldo + slot * N, v);
    }
    if (partial) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < N; j += 4) {
        s1 += (v[j] + v[j + 1]) + (v[j + 2] + v[j + 3]);
        s2 += (v[j] * v[j] + v[j + 1] * v[j + 1]) + (v[j + 2] * v[j + 2] + v[j + 3] * v[j + 3]);
      }
      a1 += (double)s1; a2 += (double)s2;
    }
  }
  if (!partial) return;
  for (int i = tid; i < 64; i += 256) reinterpret_cast<double*>(red)[i] = 0.0;
  __syncthreads();
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
    if (off < gl || off >= CL) { a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); }   // (uniform condition)
  const int wslot = (wave * 64 + lane) % CL;            // this lane's channel slot; a wave holds min(CL, 64) distinct slots
  if ((wslot % gl) == 0 && lane < CL) { red[wave][wslot / gl][0] = a1; red[wave][wslot / gl][1] = a2; }
  __syncthreads();
  if (tid < G) {
    double* o = partial + (((size_t)b * splits + sp) * G + tid) * 2;
    o[0] = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
    o[1] = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
  }
}

// The head's mixed transform on the bf16 activation stream (C = 64, whole blocks, shared matrix): out = bf16(mix_in x + mix_t T x T^T)
// with x and out bf16 NHWC.  block_transform_fast_k maps a thread to a (block, channel) pair and touches global memory 2 bytes
// at a time on a bf16 tensor (0.98 ms at 512 x 512, B = 8, against 0.24 ms on fp32); here a workgroup moves its NB blocks
// (BS x BS pixels x 64 channels each) between global memory and an LDS tile with 16-byte accesses and the threads do their
// 2-byte accesses there.  Same FMA order as block_transform_fast_k; the GroupNorm(8) partial sums are those of the stored
// (rounded) values, layout as there.
template <int BS>
__global__ __launch_bounds__(BS == 8 ? 128 : 256) void head_mix16_k(const __bf16* __restrict__ in, int ldi, __bf16* __restrict__ out, int ldo,
                                                                   int B, int H, int W, const float* __restrict__ mats, float mix_in,
                                                                   float mix_t, double* __restrict__ gnp) {
  constexpr int PX = BS * BS, NB = BS == 8 ? 2 : 4, NT = NB * 64, ROW = 72;   // LDS row: 64 channels + 8 pad (144 bytes)
  __shared__ __attribute__((aligned(16))) __bf16 tile[NB * PX * ROW];
  __shared__ float Ts[PX];
  const int tid = threadIdx.x;
  for (int i = tid; i < PX; i += NT) Ts[i] = mats[i];
  const int nby = H / BS, nbx = W / BS;
  const int64_t nblk = (int64_t)B * nby * nbx, gb0 = (int64_t)blockIdx.x * NB;
  int64_t base[NB];                                      // first pixel of each of the workgroup's blocks
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int64_t gb = gb0 + j < nblk ? gb0 + j : 0;
    const int bx = (int)(gb % nbx), by = (int)((gb / nbx) % nby), b = (int)(gb / ((int64_t)nbx * nby));
    base[j] = ((int64_t)b * H + by * BS) * W + bx * BS;
  }
  auto pixel = [&](int j, int px) -> int64_t {           // (select chain: no dynamically indexed register array)
    int64_t bj = base[0];
#pragma unroll
    for (int k = 1; k < NB; ++k) bj = j == k ? base[k] : bj;
    return bj + (int64_t)(px / BS) * W + px % BS;
  };
  // 16-byte chunk q of the workgroup: block q / (PX * 8), pixel (q / 8) % PX, channels 8 (q % 8) ...
#pragma unroll
  for (int q = tid; q < NB * PX * 8; q += NT) {
    const int blk = q / (PX * 8), px = (q >> 3) % PX, c8 = q & 7;
    if (gb0 + blk < nblk)
      *reinterpret_cast<h8_t*>(tile + (blk * PX + px) * ROW + 8 * c8) = *reinterpret_cast<const h8_t*>(in + pixel(blk, px) * ldi + 8 * c8);
  }
  __syncthreads();
  const int blk = tid >> 6, c = tid & 63;
  if (gb0 + blk < nblk) {
    __bf16* col = tile + blk * PX * ROW + c;
    float X[BS][BS], tmp[BS][BS];
#pragma unroll
    for (int u = 0; u < BS; ++u)
#pragma unroll
      for (int v = 0; v < BS; ++v) X[u][v] = (float)col[(u * BS + v) * ROW];
#pragma unroll
    for (int u = 0; u < BS; ++u) {
      float tr[BS];
#pragma unroll
      for (int k = 0; k < BS; ++k) tr[k] = Ts[u * BS + k];
#pragma unroll
      for (int v = 0; v < BS; ++v) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < BS; ++k) a = fmaf(tr[k], X[k][v], a);
        tmp[u][v] = a;
      }
    }
    float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
    for (int v = 0; v < BS; ++v) {
      float tr[BS];
#pragma unroll
      for (int k = 0; k < BS; ++k) tr[k] = Ts[v * BS + k];
#pragma unroll
      for (int u = 0; u < BS; ++u) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < BS; ++k) a = fmaf(tmp[u][k], tr[k], a);
        a = mix_in * X[u][v] + mix_t * a;
        const __bf16 h = (__bf16)a;
        col[(u * BS + v) * ROW] = h;                     // (own column: nobody else reads or writes it)
        a = (float)h;
        gs1 += a; gs2 = fmaf(a, a, gs2);
      }
    }
    if (gnp) {   // the 8 channels of a group are 8 consecutive lanes
      double d1 = (double)gs1, d2 = (double)gs2;
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
      if ((c & 7) == 0) {
        double* dst = gnp + ((gb0 + blk) * 8 + (c >> 3)) * 2;
        dst[0] = d1; dst[1] = d2;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = tid; q < NB * PX * 8; q += NT) {
    const int bq = q / (PX * 8), px = (q >> 3) % PX, c8 = q & 7;
    if (gb0 + bq < nblk)
      *reinterpret_cast<h8_t*>(out + pixel(bq, px) * ldo + 8 * c8) = *reinterpret_cast<const h8_t*>(tile + (bq * PX + px) * ROW + 8 * c8);
  }
}

__global__ void adaptive_avgpool_k(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B,
                                   int H, int W, int C4, int S) {
  const int64_t total = (int64_t)B * S * S * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c4 = i % C4;
    int64_t p = i / C4;
    int xo = p % S; p /= S;
    int yo = p % S; int b = p / S;
    int ys = (yo * H) / S, ye = ((yo + 1) * H + S - 1) / S;
    int xs = (xo * W) / S, xe = ((xo + 1) * W + S - 1) / S;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y = ys; y < ye; ++y)
      for (int x = xs; x < xe; ++x) {
        float4 v = *reinterpret_cast<const float4*>(in + (((size_t)b * H + y) * W + x) * ldi + 4 * c4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    float inv = 1.0f / (float)((ye - ys) * (xe - xs));
    *reinterpret_cast<float4*>(out + (((size_t)b * S + yo) * S + xo) * ldo + 4 * c4) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// AdaptiveAvgPool2d(s) for s in {1,2,4,8} from ONE pass over the tensor (avif_inference.py:227-234 pools the same x at
// four scales): when H and W are multiples of 8 the bins of all four scales nest, so one workgroup per (sample, 8x8 bin)
// streams its pixels (lanes over channels, the rest of the block over pixels), reduces through LDS and writes the bin
// SUM; the four pooled maps are then folded from the 64 sums.
__global__ __launch_bounds__(256) void pool8_sums_k(const float* __restrict__ in, int ldi, float* __restrict__ sums, int H,
                                                    int W, int C4) {
  __shared__ float4 red[256];
  const int bin = blockIdx.x, b = blockIdx.y, by = bin >> 3, bx = bin & 7;
  const int bh = H / 8, bw = W / 8, tid = threadIdx.x;
  const int lpp = C4 < 256 ? C4 : 256, slots = 256 / lpp, slot = tid / lpp, c4 = tid - slot * lpp;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (slot < slots)
    for (int i = slot; i < bh * bw; i += slots) {
      const int y = by * bh + i / bw, x = bx * bw + i % bw;
      const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)b * H + y) * W + x) * ldi + 4 * c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  red[tid] = acc;
  __syncthreads();
  if (tid < lpp) {
    float4 t = red[tid];
    for (int sl = 1; sl < slots; ++sl) { const float4 v = red[sl * lpp + tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    *reinterpret_cast<float4*>(sums + (((size_t)b * 64 + bin) * C4 + tid) * 4) = t;
  }
}

__global__ void pool_from8_k(const float* __restrict__ sums, float* __restrict__ out, int ldo, int B, int C4, int S, float inv) {
  const int64_t total = (int64_t)B * S * S * C4;
  const int f = 8 / S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    int64_t q = i / C4;
    const int xo = q % S; q /= S;
    const int yo = q % S; const int b = q / S;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int yy = 0; yy < f; ++yy)
      for (int xx = 0; xx < f; ++xx) {
        const float4 v = *reinterpret_cast<const float4*>(sums + (((size_t)b * 64 + (yo * f + yy) * 8 + xo * f + xx) * C4 + c4) * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    *reinterpret_cast<float4*>(out + (((size_t)b * S + yo) * S + xo) * ldo + 4 * c4) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// ------------------------------------------------------------------------------------------- block transform
// One thread = one (block, channel): Y = T X T^T with zero padding beyond H/W and cropping on store.
// The transform matrix sits in LDS (shared matrix) or is streamed per channel; X lives in registers with
// compile-time indexing only (a runtime-indexed local array would be demoted to scratch memory).
template <int BS>
__global__ __launch_bounds__(BS == 8 ? 128 : 256) void block_transform_k(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B,
                                  int H, int W, int C, const float* __restrict__ mats, int per_channel,
                                  float mix_in, float mix_t, __bf16* __restrict__ low16, __bf16* __restrict__ high16,
                                  int low) {
  __shared__ float Ts[BS * BS];
  if (!per_channel) {
    for (int i = threadIdx.x; i < BS * BS; i += blockDim.x) Ts[i] = mats[i];
    __syncthreads();
  }
  const int nby = (H + BS - 1) / BS, nbx = (W + BS - 1) / BS;
  const int64_t total = (int64_t)B * nby * nbx * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = i % C;
    int64_t p = i / C;
    int bx = p % nbx; p /= nbx;
    int by = p % nby; int b = p / nby;
    const float* T = per_channel ? mats + (size_t)c * BS * BS : Ts;
    float X[BS][BS];
#pragma unroll
    for (int u = 0; u < BS; ++u) {
      int y = by * BS + u;
#pragma unroll
      for (int v = 0; v < BS; ++v) {
        int x = bx * BS + v;
        X[u][v] = (y < H && x < W) ? in[(((size_t)b * H + y) * W + x) * ldi + c] : 0.f;
      }
    }
    // tmp = T X   (row u of tmp needs row u of T: read it once)
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
    // Y = tmp T^T : Y[u][v] = sum_k tmp[u][k] * T[v][k]
#pragma unroll
    for (int v = 0; v < BS; ++v) {
      float tr[BS];
#pragma unroll
      for (int k = 0; k < BS; ++k) tr[k] = T[v * BS + k];
      int x = bx * BS + v;
#pragma unroll
      for (int u = 0; u < BS; ++u) {
        int y = by * BS + u;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < BS; ++k) a = fmaf(tmp[u][k], tr[k], a);
        if (y < H && x < W) {
          const size_t o = (((size_t)b * H + y) * W + x);
          const float val = mix_in * X[u][v] + mix_t * a;
          out[o * ldo + c] = val;
          if (low16 && !high16) {   // one unmasked bf16 copy: operand of the fused low/high gate GEMM
            low16[o * C + c] = (__bf16)val;
          } else if (low16) {       // pre-masked bf16 operands of the separate low / high frequency gate GEMMs
            const bool lowp = is_low_freq(y, x, H, W, BS, low);
            low16[o * C + c] = (__bf16)(lowp ? val : 0.f);
            high16[o * C + c] = (__bf16)(lowp ? 0.f : val);
          }
        }
      }
    }
  }
}

// The common case of the pass above -- whole blocks (H, W multiples of BS), out = T X T^T unmixed, at most one unmasked
// bf16 copy -- without the generality: X is dead once T X exists, no bounds or mask tests.  The general kernel keeps X live
// for the runtime mix and carries the mask logic: 206 VGPRs / 2 waves per SIMD at BS = 8 and 208 us for the level-0 tensor
// (B = 16, 256 x 256 x 64) against 119 us for this mapping stand-alone (tools/micro/dct_store.hip, 5.6 TB/s; wider
// per-thread accesses -- two channels per thread, an LDS-staged patch with 16-byte I/O -- measured SLOWER there).
// gnp != null (the head's mix, C = 64): also emits the GroupNorm(8) partial sums of the OUTPUT, one (sum, sum of squares) per
// (sample, block, group) at gnp[((b * blocks_per_image + block) * 8 + group) * 2] -- the statistics pass over the mixed tensor
// (one more read of 537 MB at 512 x 512, B = 8) disappears; gn_finalize folds the slots in a fixed order.
// IO16 (bf16 activation stream, the head's mix): `in` and `out` are bf16 tensors.
template <int BS, bool PC, bool O16, bool MIX = false, bool IO16 = false>
__global__ __launch_bounds__(BS == 8 ? 128 : 256) void block_transform_fast_k(const float* __restrict__ in, int ldi, float* __restrict__ out,
                                                                             int ldo, int B, int H, int W, int C,
                                                                             const float* __restrict__ mats, __bf16* __restrict__ o16,
                                                                             float mix_in, float mix_t, double* __restrict__ gnp) {
  __shared__ float Ts[BS * BS];
  if (!PC) {
    for (int i = threadIdx.x; i < BS * BS; i += blockDim.x) Ts[i] = mats[i];
    __syncthreads();
  }
  const int nby = H / BS, nbx = W / BS;
  const int64_t total = (int64_t)B * nby * nbx * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int bx = (int)(p % nbx); p /= nbx;
    const int by = (int)(p % nby);
    const int b = (int)(p / nby);
    const float* T = PC ? mats + (size_t)c * BS * BS : Ts;
    const size_t base = (((size_t)b * H + by * BS) * W + bx * BS);
    float gs1 = 0.f, gs2 = 0.f;
    float X[BS][BS];
#pragma unroll
    for (int u = 0; u < BS; ++u)
#pragma unroll
      for (int v = 0; v < BS; ++v)
        X[u][v] = IO16 ? (float)reinterpret_cast<const __bf16*>(in)[(base + (size_t)u * W + v) * ldi + c] : in[(base + (size_t)u * W + v) * ldi + c];
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
        const size_t o = base + (size_t)u * W + v;
        if (MIX) a = mix_in * X[u][v] + mix_t * a;       // the head's u5 + 0.1 DCT(u5) (webp_inference.py:396)
        if (IO16) { const __bf16 h = (__bf16)a; reinterpret_cast<__bf16*>(out)[o * ldo + c] = h; a = (float)h; }   // (sums of what is stored)
        else out[o * ldo + c] = a;
        if (O16) o16[o * C + c] = (__bf16)a;
        if (MIX) { gs1 += a; gs2 = fmaf(a, a, gs2); }
      }
    }
    if (MIX && gnp) {   // C = 64 (host check): the 8 channels of a group are 8 consecutive lanes
      double d1 = (double)gs1, d2 = (double)gs2;
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
      if ((c & 7) == 0) {
        double* dst = gnp + ((((size_t)b * nby + by) * nbx + bx) * 8 + (c >> 3)) * 2;
        dst[0] = d1; dst[1] = d2;
      }
    }
  }
}

__global__ void mul4_add_k(const float* base, int ldb, const float* a, int lda, const float* b, int ldb2,
                           const float* c, int ldc, const float* d, int ldd, float* out, int ldo, int64_t M, int C4) {
  const int64_t total = M * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c4 = i % C4;
    int64_t m = i / C4;
    float4 v = *reinterpret_cast<const float4*>(a + m * lda + 4 * c4);
    if (b) { float4 w = *reinterpret_cast<const float4*>(b + m * ldb2 + 4 * c4); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
    if (c) { float4 w = *reinterpret_cast<const float4*>(c + m * ldc + 4 * c4); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
    if (d) { float4 w = *reinterpret_cast<const float4*>(d + m * ldd + 4 * c4); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
    if (base) { float4 w = *reinterpret_cast<const float4*>(base + m * ldb + 4 * c4); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    *reinterpret_cast<float4*>(out + m * ldo + 4 * c4) = v;
  }
}

// AVIF frequency block tail (avif_inference.py:227-256): out = base + xt * avg * col * edge, where
// avg = 0.25 * (up(g1) + up(g2) + up(g4) + up(g8)) and up() is F.interpolate(size=(H, W), bilinear, align_corners=False)
// of the S x S pooled gates.  The four tiny gate maps are interpolated on the fly (same summation order as the
// accumulate-then-scale passes this replaces), so the full-resolution `avg` tensor is never written or re-read.
__global__ void avif_combine_k(const float* __restrict__ base, const float* __restrict__ xt, const float* __restrict__ gates,
                               const float* __restrict__ col, const float* __restrict__ edge, float* __restrict__ out, int B,
                               int H, int W, int C4) {
  const int64_t total = (int64_t)B * H * W * C4;
  const int C = C4 * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    int64_t m = i / C4;
    int64_t q = m;
    const int xo = q % W; q /= W;
    const int yo = q % H; const int b = q / H;
    float4 avg = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t goff = 0;                                   // gate maps are stored back to back: [B][S][S][C] for S = 1, 2, 4, 8
#pragma unroll
    for (int si = 0; si < 4; ++si) {
      const int S = 1 << si;
      int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
      lin_coord(yo, S, (float)S / (float)H, y0, y1, wy0, wy1);
      lin_coord(xo, S, (float)S / (float)W, x0, x1, wx0, wx1);
      const float* ib = gates + goff + (size_t)b * S * S * C + 4 * c4;
      const float4 v00 = *reinterpret_cast<const float4*>(ib + ((size_t)y0 * S + x0) * C);
      const float4 v01 = *reinterpret_cast<const float4*>(ib + ((size_t)y0 * S + x1) * C);
      const float4 v10 = *reinterpret_cast<const float4*>(ib + ((size_t)y1 * S + x0) * C);
      const float4 v11 = *reinterpret_cast<const float4*>(ib + ((size_t)y1 * S + x1) * C);
      avg.x += wy0 * (wx0 * v00.x + wx1 * v01.x) + wy1 * (wx0 * v10.x + wx1 * v11.x);
      avg.y += wy0 * (wx0 * v00.y + wx1 * v01.y) + wy1 * (wx0 * v10.y + wx1 * v11.y);
      avg.z += wy0 * (wx0 * v00.z + wx1 * v01.z) + wy1 * (wx0 * v10.z + wx1 * v11.z);
      avg.w += wy0 * (wx0 * v00.w + wx1 * v01.w) + wy1 * (wx0 * v10.w + wx1 * v11.w);
      goff += (size_t)B * S * S * C;
    }
    float4 v = *reinterpret_cast<const float4*>(xt + m * C + 4 * c4);
    const float4 cw = *reinterpret_cast<const float4*>(col + m * C + 4 * c4);
    const float4 ew = *reinterpret_cast<const float4*>(edge + m * C + 4 * c4);
    const float4 bw = *reinterpret_cast<const float4*>(base + m * C + 4 * c4);
    v.x = v.x * (avg.x * 0.25f) * cw.x * ew.x + bw.x;
    v.y = v.y * (avg.y * 0.25f) * cw.y * ew.y + bw.y;
    v.z = v.z * (avg.z * 0.25f) * cw.z * ew.z + bw.z;
    v.w = v.w * (avg.w * 0.25f) * cw.w * ew.w + bw.w;
    *reinterpret_cast<float4*>(out + m * C + 4 * c4) = v;
  }
}

// ------------------------------------------------------------------------------------------- time embedding
// sinusoid -> Linear(256,1024) -> SiLU -> Linear(1024,256) (webp_inference.py:145-151), two launches: one wave per output
// unit streams its weight row once (coalesced float4 per lane) and forms the dot product for every sample of the batch
// (the one-workgroup-per-sample version walked 1 KiB-strided rows per thread: 72 us per forward).
__global__ __launch_bounds__(256) void time_embed1_k(const float* __restrict__ t, TimeEmbedW w, int B, float* __restrict__ hidden) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x * 4 + wave;                      // hidden unit 0..1023
  const float4 wv = *reinterpret_cast<const float4*>(w.w0 + (size_t)n * 256 + 4 * lane);
  const float coef = (float)(-9.210340371976184 / 127.0);   // -ln(10000)/(half-1), rounded to fp32 like torch
  float f[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = expf((float)((4 * lane + j) & 127) * coef);
  const float bias = w.b0[n];
  for (int b = 0; b < B; ++b) {
    const float tb = t[b];
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float a = tb * f[j]; e[j] = (4 * lane + j) < 128 ? sinf(a) : cosf(a); }
    float acc = fmaf(wv.x, e[0], fmaf(wv.y, e[1], fmaf(wv.z, e[2], wv.w * e[3])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
      acc += bias;
      hidden[(size_t)b * 1024 + n] = acc / (1.0f + __expf(-acc));
    }
  }
}

__global__ __launch_bounds__(256) void time_embed2_k(const float* __restrict__ hidden, TimeEmbedW w, int B, float* __restrict__ temb) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x * 4 + wave;                      // output 0..255
  float4 wv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const float4*>(w.w2 + (size_t)n * 1024 + q * 256 + 4 * lane);
  const float bias = w.b2[n];
  for (int b = 0; b < B; ++b) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 h = *reinterpret_cast<const float4*>(hidden + (size_t)b * 1024 + q * 256 + 4 * lane);
      acc = fmaf(wv[q].x, h.x, acc); acc = fmaf(wv[q].y, h.y, acc); acc = fmaf(wv[q].z, h.z, acc); acc = fmaf(wv[q].w, h.w, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) temb[(size_t)b * 256 + n] = acc + bias;
  }
}

// all 13 time_proj Linear(256, out_c) at once: tproj[b][n] = wt[n] . temb[b] + bt[n]
__global__ __launch_bounds__(256) void time_proj_k(const float* __restrict__ temb, TimeEmbedW w,
                                                   float* __restrict__ tproj) {
  __shared__ float e[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  e[tid] = temb[(size_t)b * 256 + tid];
  __syncthreads();
  const int n = blockIdx.x * 256 + tid;
  if (n >= w.NT) return;
  const float* wr = w.wt + (size_t)n * 256;
  float acc = 0.f;
  for (int k = 0; k < 256; k += 4) {
    float4 wv = *reinterpret_cast<const float4*>(wr + k);
    acc = fmaf(wv.x, e[k], acc); acc = fmaf(wv.y, e[k + 1], acc);
    acc = fmaf(wv.z, e[k + 2], acc); acc = fmaf(wv.w, e[k + 3], acc);
  }
  tproj[(size_t)b * w.NT + n] = acc + w.bt[n];
}

__global__ void cl_boost_k(const float* cl, int B, float a0, float a1, float lo, float hi, float* out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = fminf(fmaxf(a0 + a1 * (1.0f - cl[b]), lo), hi);
}

__global__ void nchw3_to_nhwc4_k(const float* __restrict__ in, float* __restrict__ out, int B, int HW) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = i / HW, p = i - b * HW;
    const float* ip = in + b * 3 * HW + p;
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(ip[0], ip[HW], ip[2 * (int64_t)HW], 0.f);
  }
}

__global__ void nhwc4_to_nchw3_tanh_k(const float* __restrict__ in, float* __restrict__ out, int B, int HW) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = i / HW, p = i - b * HW;
    float4 v = *reinterpret_cast<const float4*>(in + i * 4);
    out[(b * 3 + 0) * HW + p] = tanhf(v.x);
    out[(b * 3 + 1) * HW + p] = tanhf(v.y);
    out[(b * 3 + 2) * HW + p] = tanhf(v.z);
  }
}

inline unsigned grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)g;
}

}  // namespace

// upper bound of the number of splits any launch uses (sizes the partial buffer: B * G * splits * 2 doubles)
int gn_splits(int HW, int C) {
  (void)C;
  int s = cdiv(HW, 64);
  if (s < 1) s = 1;
  if (s > GN_MAX_SPLITS) s = GN_MAX_SPLITS;
  return s;
}

void launch_gn_scale_shift(const float* x, int ldx, int B, int HW, int C, int G, const float* gamma,
                           const float* beta, double* partial, float* scale, float* shift, hipStream_t stream, int ldsc, int zero_to) {
  // scale/shift rows are padded to a multiple of 4 channels (zeros) so igemm can read float4
  const int Cs = ldsc > 0 ? ldsc : round_up(C, 4);
  if (zero_to < 0) zero_to = ldsc > 0 ? 0 : Cs;
  if (C > 1024) throw std::runtime_error("group norm: C > 1024 unsupported");
  const int cpg = C / G;
  const bool by_group = cpg >= 32 && (cpg % 4) == 0 && cpg <= 1024 && (ldx % 4) == 0;
  int splits = cdiv(HW, by_group ? 64 : GN_PIX_PER_SPLIT);
  splits = splits < 1 ? 1 : (splits > GN_MAX_SPLITS ? GN_MAX_SPLITS : splits);
  if (by_group)
    hipLaunchKernelGGL(gn_partial_group, dim3(G * splits, B), dim3(256), 0, stream, x, ldx, HW, C, G, splits, partial);
  else
    hipLaunchKernelGGL(gn_partial, dim3(splits, B), dim3(256), 0, stream, x, ldx, HW, C, G, splits, partial);
  hipLaunchKernelGGL(gn_finalize, dim3(G, B), dim3(256), 0, stream, partial, splits, (double)HW * (C / G), C, Cs, G,
                     gamma, beta, scale, shift, zero_to);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_gn_from_sums(const double* sums, int B, int HW, int C, int G, const float* gamma, const float* beta, float* scale,
                         float* shift, hipStream_t stream, int ldsc, int zero_to) {
  const int Cs = ldsc > 0 ? ldsc : round_up(C, 4);
  hipLaunchKernelGGL(gn_finalize, dim3(G, B), dim3(256), 0, stream, sums, HW / 32, (double)HW * (C / G), C, Cs, G, gamma, beta,
                     scale, shift, ldsc > 0 ? zero_to : Cs);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_maxpool2(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s) {
  int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 4);
  if (total <= 0) return;
  hipLaunchKernelGGL(maxpool2_k, dim3(grid_for(total)), dim3(256), 0, s, in, ldi, out, ldo, B, H, W, C / 4);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_bilinear(const float* in, int ldi, float* out, int ldo, int B, int Hi, int Wi, int Ho, int Wo, int C,
                     float scale_h, float scale_w, int accumulate, float out_scale, hipStream_t s) {
  int64_t total = (int64_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(bilinear_k, dim3(grid_for(total)), dim3(256), 0, s, in, ldi, out, ldo, B, Hi, Wi, Ho, Wo, C / 4,
                     scale_h, scale_w, accumulate, out_scale);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_upsample2(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s) {
  launch_bilinear(in, ldi, out, ldo, B, H, W, 2 * H, 2 * W, C, 0.5f, 0.5f, 0, 1.0f, s);
}

// pool / upsample / statistics passes of the inter-block stream (stream_k): returns the number of splits per sample the partial
// sums were written with (what launch_gn_from_partials needs), 0 without statistics
namespace {
template <int MODE, typename T>
int launch_stream(const void* in, int ldi, void* out, int ldo, int B, int Hi, int Wi, int C, int G, double* partial, hipStream_t s) {
  constexpr int N = LaneVec<T>::N;
  const int CL = C / N;
  const bool pow2 = CL > 0 && (CL & (CL - 1)) == 0;
  if ((C % N) != 0 || !pow2 || CL > 256 || (ldi % N) != 0 || (MODE != S16_STATS && (ldo % N) != 0) || (MODE == S16_POOL && ((Hi | Wi) & 1)))
    throw std::runtime_error("stream pass: channel count must be 16 bytes * 2^k (at most 256 lanes per pixel) with 16-byte aligned rows");
  int gl = 1;
  if (partial) {
    if (!stream_stats_supported(C, G, (int)sizeof(T))) throw std::runtime_error("stream pass: a group must be 16 bytes * 2^k of channels (at most 64 lanes of 16 bytes), at most 8 groups");
    gl = C / G / N;
  }
  const int Ho = MODE == S16_POOL ? Hi / 2 : MODE == S16_UP ? 2 * Hi : Hi, Wo = MODE == S16_POOL ? Wi / 2 : MODE == S16_UP ? 2 * Wi : Wi;
  const int ppi = 256 / CL;
  int splits = cdiv(Ho * Wo, ppi * 4);
  splits = splits < 1 ? 1 : (splits > GN_MAX_SPLITS ? GN_MAX_SPLITS : splits);
  if (Ho * Wo <= 0 || B <= 0) return 0;
  hipLaunchKernelGGL((stream_k<MODE, T>), dim3(splits, B), dim3(256), 0, s, reinterpret_cast<const T*>(in), ldi, reinterpret_cast<T*>(out), ldo,
                     Hi, Wi, CL, gl, G, splits, FastDiv((unsigned)Wo), partial);
  DDPMIR_HIP_CHECK(hipGetLastError());
  return partial ? splits : 0;
}
}  // namespace

bool stream_stats_supported(int C, int G, int elem_bytes) {
  const int N = 16 / elem_bytes;
  if (G < 1 || G > 8 || (C % G) != 0 || ((C / G) % N) != 0 || (C % N) != 0) return false;
  const int gl = C / G / N, CL = C / N;
  // gl <= 64: stream_k folds a group inside one wave (lanes with wslot % gl == 0 publish); a group spanning several waves
  // (e.g. C = 1024, G = 1 in bf16) would lose the waves that hold slots 64 .. gl-1 -- not a shape the model has, so it is rejected
  return (gl & (gl - 1)) == 0 && (CL & (CL - 1)) == 0 && CL <= 256 && gl <= 64;
}
int launch_stats16(const void* x16, int ldx, int B, int H, int W, int C, int G, double* partial, hipStream_t s) {
  return launch_stream<S16_STATS, __bf16>(x16, ldx, nullptr, 0, B, H, W, C, G, partial, s);
}
int launch_maxpool2_16(const void* in16, int ldi, void* out16, int ldo, int B, int H, int W, int C, hipStream_t s, double* partial, int G) {
  return launch_stream<S16_POOL, __bf16>(in16, ldi, out16, ldo, B, H, W, C, G, partial, s);
}
int launch_upsample2_16(const void* in16, int ldi, void* out16, int ldo, int B, int H, int W, int C, hipStream_t s, double* partial, int G) {
  return launch_stream<S16_UP, __bf16>(in16, ldi, out16, ldo, B, H, W, C, G, partial, s);
}
int launch_maxpool2_stats(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s, double* partial, int G) {
  return launch_stream<S16_POOL, float>(in, ldi, out, ldo, B, H, W, C, G, partial, s);
}
int launch_upsample2_stats(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s, double* partial, int G) {
  return launch_stream<S16_UP, float>(in, ldi, out, ldo, B, H, W, C, G, partial, s);
}

void launch_adaptive_avgpool(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, int S,
                             hipStream_t s) {
  int64_t total = (int64_t)B * S * S * (C / 4);
  hipLaunchKernelGGL(adaptive_avgpool_k, dim3(grid_for(total)), dim3(256), 0, s, in, ldi, out, ldo, B, H, W, C / 4, S);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

bool pool8_supported(int H, int W, int C) { return (H % 8) == 0 && (W % 8) == 0 && H * W >= 256 && (C % 4) == 0 && C <= 1024; }

void launch_pool8_sums(const float* in, int ldi, float* sums, int B, int H, int W, int C, hipStream_t s) {
  hipLaunchKernelGGL(pool8_sums_k, dim3(64, B), dim3(256), 0, s, in, ldi, sums, H, W, C / 4);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_pool_from8(const float* sums, float* out, int ldo, int B, int H, int W, int C, int S, hipStream_t s) {
  const int64_t total = (int64_t)B * S * S * (C / 4);
  hipLaunchKernelGGL(pool_from8_k, dim3(grid_for(total)), dim3(256), 0, s, sums, out, ldo, B, C / 4, S,
                     1.0f / (float)((H / S) * (W / S)));
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_block_transform(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, int bs,
                            const float* mats, int per_channel, float mix_in, float mix_t, hipStream_t s,
                            void* low16, void* high16, int low, double* gn_partial_out, int io16) {
  int64_t total = (int64_t)B * cdiv(H, bs) * cdiv(W, bs) * C;
  if (io16 && !((H % bs) == 0 && (W % bs) == 0 && !high16 && !low16 && !per_channel && (bs == 4 || bs == 8) && !(mix_in == 0.f && mix_t == 1.f)))
    throw std::runtime_error("block transform: bf16 in / out is the head's mixed transform on whole blocks only");
  __bf16* l16 = reinterpret_cast<__bf16*>(low16);
  __bf16* h16 = reinterpret_cast<__bf16*>(high16);
  if ((H % bs) == 0 && (W % bs) == 0 && !h16 && (bs == 4 || bs == 8)) {
    auto go = [&](auto kern, int block) {
      hipLaunchKernelGGL(kern, dim3(grid_for(total, block)), dim3(block), 0, s, in, ldi, out, ldo, B, H, W, C, mats, l16, mix_in, mix_t,
                         gn_partial_out);
    };
    if (gn_partial_out && (C != 64 || mix_in == 0.f)) throw std::runtime_error("block transform: fused GroupNorm sums need the 64-channel mixed output");
    if (!(mix_in == 0.f && mix_t == 1.f)) {   // mixed output: the model's head only (shared matrix, no bf16 copy)
      if (per_channel || l16) goto general;
      if (io16 && C == 64 && (ldi % 8) == 0 && (ldo % 8) == 0) {   // LDS-staged 16-byte I/O (head_mix16_k)
        const int64_t nblk = (int64_t)B * (H / bs) * (W / bs);
        const __bf16* i16 = reinterpret_cast<const __bf16*>(in);
        __bf16* o16 = reinterpret_cast<__bf16*>(out);
        if (bs == 4) hipLaunchKernelGGL(head_mix16_k<4>, dim3((unsigned)cdiv64(nblk, 4)), dim3(256), 0, s, i16, ldi, o16, ldo, B, H, W, mats, mix_in, mix_t, gn_partial_out);
        else hipLaunchKernelGGL(head_mix16_k<8>, dim3((unsigned)cdiv64(nblk, 2)), dim3(128), 0, s, i16, ldi, o16, ldo, B, H, W, mats, mix_in, mix_t, gn_partial_out);
      }
      else if (io16) { if (bs == 4) go(block_transform_fast_k<4, false, false, true, true>, 256); else go(block_transform_fast_k<8, false, false, true, true>, 128); }
      else if (bs == 4) go(block_transform_fast_k<4, false, false, true>, 256);
      else go(block_transform_fast_k<8, false, false, true>, 128);
    } else if (bs == 4) {
      if (per_channel) { if (l16) go(block_transform_fast_k<4, true, true>, 256); else go(block_transform_fast_k<4, true, false>, 256); }
      else { if (l16) go(block_transform_fast_k<4, false, true>, 256); else go(block_transform_fast_k<4, false, false>, 256); }
    } else {
      if (per_channel) { if (l16) go(block_transform_fast_k<8, true, true>, 128); else go(block_transform_fast_k<8, true, false>, 128); }
      else { if (l16) go(block_transform_fast_k<8, false, true>, 128); else go(block_transform_fast_k<8, false, false>, 128); }
    }
    DDPMIR_HIP_CHECK(hipGetLastError());
    return;
  }
general:
  if (bs == 4)
    hipLaunchKernelGGL(block_transform_k<4>, dim3(grid_for(total)), dim3(256), 0, s, in, ldi, out, ldo, B, H, W, C,
                       mats, per_channel, mix_in, mix_t, l16, h16, low);
  else if (bs == 8)
    hipLaunchKernelGGL(block_transform_k<8>, dim3(grid_for(total, 128)), dim3(128), 0, s, in, ldi, out, ldo, B, H, W,
                       C, mats, per_channel, mix_in, mix_t, l16, h16, low);
  else
    throw std::runtime_error("block transform: block size must be 4 or 8");
  if (gn_partial_out) throw std::runtime_error("block transform: fused GroupNorm sums need whole blocks and a shared matrix");
  DDPMIR_HIP_CHECK(hipGetLastError());
}

bool block_transform_gn_fusable(int H, int W, int C, int bs) { return C == 64 && (bs == 4 || bs == 8) && (H % bs) == 0 && (W % bs) == 0; }

void launch_gn_from_partials(const double* partial, int splits, int B, double n_per_group, int C, int G, const float* gamma,
                             const float* beta, float* scale, float* shift, hipStream_t stream, int ldsc, int zero_to) {
  const int Cs = ldsc > 0 ? ldsc : round_up(C, 4);
  hipLaunchKernelGGL(gn_finalize, dim3(G, B), dim3(256), 0, stream, partial, splits, n_per_group, C, Cs, G, gamma, beta,
                     scale, shift, ldsc > 0 ? (zero_to < 0 ? 0 : zero_to) : Cs);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_avif_combine(const float* base, const float* xt, const float* gates, const float* col, const float* edge,
                         float* out, int B, int H, int W, int C, hipStream_t s) {
  const int64_t total = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(avif_combine_k, dim3(grid_for(total)), dim3(256), 0, s, base, xt, gates, col, edge, out, B, H, W, C / 4);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_mul4_add(const float* base, int ldb, const float* a, int lda, const float* b, int ldb2, const float* c,
                     int ldc, const float* d, int ldd, float* out, int ldo, int64_t M, int C, hipStream_t s) {
  int64_t total = M * (C / 4);
  hipLaunchKernelGGL(mul4_add_k, dim3(grid_for(total)), dim3(256), 0, s, base, ldb, a, lda, b, ldb2, c, ldc, d, ldd,
                     out, ldo, M, C / 4);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_time_embed(const float* t, int B, const TimeEmbedW& w, float* hidden, float* temb, float* tproj,
                       hipStream_t s) {
  hipLaunchKernelGGL(time_embed1_k, dim3(256), dim3(256), 0, s, t, w, B, hidden);
  hipLaunchKernelGGL(time_embed2_k, dim3(64), dim3(256), 0, s, hidden, w, B, temb);
  hipLaunchKernelGGL(time_proj_k, dim3(cdiv(w.NT, 256), B), dim3(256), 0, s, temb, w, tproj);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_cl_boost(const float* cl, int B, float a0, float a1, float lo, float hi, float* out, hipStream_t s) {
  hipLaunchKernelGGL(cl_boost_k, dim3(cdiv(B, 64)), dim3(64), 0, s, cl, B, a0, a1, lo, hi, out);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_nhwc4_to_nchw3_tanh(const float* in, float* out, int B, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(nhwc4_to_nchw3_tanh_k, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, s, in, out, B, H * W);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_nchw3_to_nhwc4(const float* in, float* out, int B, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(nchw3_to_nhwc4_k, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, s, in, out, B, H * W);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

// Per-step kernels of the DDRM / DDPM sampling loops (reference: webp_inference.py:557-602).
// Arithmetic that feeds the uint8 truncation is written with explicit round-to-nearest intrinsics so that
// hipcc cannot contract a*b+c into an fma: torch evaluates x*127.5+127.5 as two rounded fp32 ops, and a
// 1-ulp difference there flips a pixel by 1/127.5 after the codec.
#include "kernels.h"

// hipcc defaults to -ffp-contract=fast and __fmul_rn/__fadd_rn are plain operators in the HIP headers, so
// contraction has to be switched off for this translation unit (build.py also passes -ffp-contract=off).
#pragma clang fp contract(off)

namespace ddpmir {

namespace {

__global__ void quantize_u8_k(const float* __restrict__ x, uint8_t* __restrict__ u8, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = __fadd_rn(__fmul_rn(x[i], 127.5f), 127.5f);
    v = fminf(fmaxf(v, 0.f), 255.f);
    u8[i] = (uint8_t)(int)v;   // truncation, like .to(torch.uint8)
  }
}

// x' = (x_theta - dec) + y ; x_t = (eta_b*x' + (1-eta_b)*x_theta) + eta*(noise*(t*coeff))
__global__ void ddrm_update_k(const float* __restrict__ x_theta, const uint8_t* __restrict__ dec_u8,
                              const float* __restrict__ y, const float* __restrict__ noise, float* __restrict__ x_out,
                              int64_t n, float eta, float eta_b, float one_minus_eta_b, float noise_scale, int last) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float dec = __fmul_rn(__fsub_rn(__fdiv_rn((float)dec_u8[i], 255.0f), 0.5f), 2.0f);
    float th = x_theta[i];
    float xp = __fadd_rn(__fsub_rn(th, dec), y[i]);
    if (last) {
      x_out[i] = xp;
    } else {
      float rnd = __fmul_rn(noise[i], noise_scale);
      float mix = __fadd_rn(__fmul_rn(eta_b, xp), __fmul_rn(one_minus_eta_b, th));
      x_out[i] = __fadd_rn(mix, __fmul_rn(eta, rnd));
    }
  }
}

__global__ void fill_t_k(float* t, int B, float v) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) t[b] = v;
}

// ---- per-timestep scalars in device memory (StepParams, kernels.h): the kernels of one timestep read the loop index and
// everything derived from it from there, so the captured timestep graph is replayed unchanged for every i.
// t = float(i)/steps and noise_scale = t*coeff are single correctly-rounded fp32 operations, exactly what the host
// formed before (webp_inference.py:568,588).
__device__ __forceinline__ void step_params_derive(StepParams* p) {
  p->t = __fdiv_rn((float)p->i, (float)p->steps);
  p->noise_scale = __fmul_rn(p->t, p->coeff);
  p->last = p->i == 0 ? 1 : 0;
  p->seed = p->seed_base + (unsigned long long)p->i * 0x9E3779B97F4A7C15ull;
}
__global__ void step_params_set_k(StepParams* p, int i, int steps, float coeff, unsigned long long seed_base,
                                  unsigned long long rng_offset, const float* noise, long long noise_stride) {
  p->i = i; p->steps = steps; p->coeff = coeff; p->seed_base = seed_base; p->rng_offset = rng_offset;
  p->noise = noise; p->noise_stride = noise_stride;
  step_params_derive(p);
}
__global__ void step_params_advance_k(StepParams* p) {   // last node of a timestep: i -> i - 1
  p->i -= 1;
  step_params_derive(p);
}
__global__ void fill_t_p_k(float* t, int B, const StepParams* __restrict__ p) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) t[b] = p->t;
}
// DDRM update with the step scalars (and, for injected noise, the [steps][...] base pointer) read from *p
__global__ void ddrm_update_p_k(const float* __restrict__ x_theta, const uint8_t* __restrict__ dec_u8,
                                const float* __restrict__ y, const float* __restrict__ dev_noise, long long lane_off,
                                float* __restrict__ x_out, int64_t n, float eta, float eta_b, float one_minus_eta_b,
                                const StepParams* __restrict__ p) {
  const int last = p->last;
  const float noise_scale = p->noise_scale;
  const float* noise = p->noise ? p->noise + (long long)p->i * p->noise_stride + lane_off : dev_noise;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float dec = __fmul_rn(__fsub_rn(__fdiv_rn((float)dec_u8[i], 255.0f), 0.5f), 2.0f);
    float th = x_theta[i];
    float xp = __fadd_rn(__fsub_rn(th, dec), y[i]);
    if (last) {
      x_out[i] = xp;
    } else {
      float rnd = __fmul_rn(noise[i], noise_scale);
      float mix = __fadd_rn(__fmul_rn(eta_b, xp), __fmul_rn(one_minus_eta_b, th));
      x_out[i] = __fadd_rn(mix, __fmul_rn(eta, rnd));
    }
  }
}

// ------------------------------------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ void randn_k(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset, const StepParams* __restrict__ sp) {
  if (sp) {   // timestep form: key and counter base from the device-side step scalars; nothing to draw at the last step
    if (sp->last || sp->noise) return;                                   // ... or with injected noise
    seed = sp->seed; offset = sp->rng_offset;
  }
  const int64_t nq = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    uint64_t ctr = offset + (uint64_t)q;
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float inv = 2.3283064365386963e-10f;   // 2^-32
    float u0 = ((float)c0 + 0.5f) * inv, u1 = ((float)c1 + 0.5f) * inv;
    float u2 = ((float)c2 + 0.5f) * inv, u3 = ((float)c3 + 0.5f) * inv;
    u0 = fminf(fmaxf(u0, 1e-12f), 1.0f); u2 = fminf(fmaxf(u2, 1e-12f), 1.0f);
    float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float z[4];
    z[0] = r0 * cosf(6.283185307179586f * u1); z[1] = r0 * sinf(6.283185307179586f * u1);
    z[2] = r1 * cosf(6.283185307179586f * u3); z[3] = r1 * sinf(6.283185307179586f * u3);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * q + j < n) out[4 * q + j] = z[j];
  }
}

// ------------------------------------------------------------------------------------------- separable DFT
// rows:  T[p][y][v] = sum_x in[p][y][x] * w^(v*x),  w = exp(-+2 pi i / W)  (complex in optional)
// One workgroup per RB = 8 consecutive rows; the rows sit in LDS interleaved [x][8] so one x of all 8 rows is two
// broadcast ds_read_b128 per component, and the per-lane twiddle read is shared by the 8 rows.  Each thread produces
// output bins v = tid, tid+256...  Twiddles come from an n-entry table indexed by (v*x) mod n -- exact periodicity,
// no drift.  The per-output summation order (x ascending, mul/mul/sub/add) is fixed: results do not depend on RB.
constexpr int DFT_RB = 8;
__global__ __launch_bounds__(256) void dft_rows_k(const float* __restrict__ in_re, const float* __restrict__ in_im,
                                                  float* __restrict__ out_re, float* __restrict__ out_im, int W,
                                                  int64_t nrows, const float* __restrict__ cs,
                                                  const float* __restrict__ sn, float sign, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* xr = sh; float* xi = sh + DFT_RB * W; float* tc = sh + 2 * DFT_RB * W; float* ts = tc + W;
  const int64_t row0 = (int64_t)blockIdx.x * DFT_RB;
  for (int i = threadIdx.x; i < DFT_RB * W; i += 256) {
    const int r = i / W, x = i - r * W;
    const bool ok = row0 + r < nrows;
    xr[x * DFT_RB + r] = ok ? in_re[(row0 + r) * W + x] : 0.f;
    xi[x * DFT_RB + r] = (ok && in_im) ? in_im[(row0 + r) * W + x] : 0.f;
  }
  for (int x = threadIdx.x; x < W; x += 256) { tc[x] = cs[x]; ts[x] = sign * sn[x]; }
  __syncthreads();
  for (int v = threadIdx.x; v < W; v += 256) {
    float ar[DFT_RB], ai[DFT_RB];
#pragma unroll
    for (int r = 0; r < DFT_RB; ++r) { ar[r] = 0.f; ai[r] = 0.f; }
    int idx = 0;
    for (int x = 0; x < W; ++x) {
      const float c = tc[idx], sv = ts[idx];       // e^{i*sign*2pi*v*x/W} = c + i s
      float a[DFT_RB], b[DFT_RB];
      *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(xr + x * DFT_RB);
      *reinterpret_cast<float4*>(a + 4) = *reinterpret_cast<const float4*>(xr + x * DFT_RB + 4);
      *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(xi + x * DFT_RB);
      *reinterpret_cast<float4*>(b + 4) = *reinterpret_cast<const float4*>(xi + x * DFT_RB + 4);
#pragma unroll
      for (int r = 0; r < DFT_RB; ++r) {
        ar[r] += a[r] * c - b[r] * sv;
        ai[r] += a[r] * sv + b[r] * c;
      }
      idx += v; if (idx >= W) idx -= W;
    }
#pragma unroll
    for (int r = 0; r < DFT_RB; ++r)
      if (row0 + r < nrows) {
        out_re[(row0 + r) * W + v] = ar[r] * scale;
        out_im[(row0 + r) * W + v] = ai[r] * scale;
      }
  }
}

// columns: O[p][u][x] = sum_y in[p][y][x] * w^(u*y).  Workgroup = (plane p, 64-wide x strip, block of 4*UB output
// rows); lane = x (coalesced), every wave owns UB = 8 output rows u, so one pair of coalesced loads per y feeds 8
// complex MACs and the twiddles are wave-uniform (broadcast LDS reads).  Same per-output summation order as above.
constexpr int DFT_UB = 8;
__global__ __launch_bounds__(256) void dft_cols_k(const float* __restrict__ in_re, const float* __restrict__ in_im,
                                                  float* __restrict__ out_re, float* __restrict__ out_im, int H, int W,
                                                  const float* __restrict__ cs, const float* __restrict__ sn,
                                                  float sign, float scale, int real_only) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* tc = sh; float* ts = sh + H;
  for (int y = threadIdx.x; y < H; y += 256) { tc[y] = cs[y]; ts[y] = sign * sn[y]; }
  __syncthreads();
  const int p = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int u0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * DFT_UB;
  if (x >= W || u0 >= H) return;
  const float* br = in_re + (size_t)p * H * W + x;
  const float* bi = in_im + (size_t)p * H * W + x;
  float ar[DFT_UB], ai[DFT_UB];
  int idx[DFT_UB], inc[DFT_UB];
#pragma unroll
  for (int j = 0; j < DFT_UB; ++j) { ar[j] = 0.f; ai[j] = 0.f; idx[j] = 0; inc[j] = u0 + j < H ? u0 + j : 0; }
  for (int y = 0; y < H; ++y) {
    const float vr = br[(size_t)y * W], vi = bi[(size_t)y * W];
#pragma unroll
    for (int j = 0; j < DFT_UB; ++j) {
      const float c = tc[idx[j]], sv = ts[idx[j]];
      ar[j] += vr * c - vi * sv;
      ai[j] += vr * sv + vi * c;
      idx[j] += inc[j]; if (idx[j] >= H) idx[j] -= H;
    }
  }
#pragma unroll
  for (int j = 0; j < DFT_UB; ++j) {
    const int u = u0 + j;
    if (u >= H) break;
    out_re[((size_t)p * H + u) * W + x] = ar[j] * scale;
    if (!real_only) out_im[((size_t)p * H + u) * W + x] = ai[j] * scale;
  }
}

// ---- arg(fft2(ref)) in double precision.  The phase of a bin whose magnitude sits at rounding level (a JPEG-decoded y can
// have a spectrum bin that is zero up to 1e-7) is decided by the error of the transform, not by the data: an fp32 direct
// DFT (error ~ N eps) then disagrees with the reference's FFT (error ~ log N eps) about the sign of such a bin and the image
// leaves the reference trajectory at its first phase fix.  This transform runs once per sample() call, so it is simply
// done in double: rows, then columns fused with the normalisation to the unit phase (fp32 out).
// rows: one workgroup per RB64 = 4 rows; the rows and the twiddle tables sit in LDS (doubles), thread = output bin v
constexpr int DFT64_RB = 4;
__global__ __launch_bounds__(256) void dft_rows_f64_k(const float* __restrict__ in, double* __restrict__ out_re, double* __restrict__ out_im, int W,
                                                      int64_t nrows, const double* __restrict__ cs, const double* __restrict__ sn) {
  extern __shared__ __attribute__((aligned(16))) double sh64[];
  double* xr = sh64; double* tc = sh64 + DFT64_RB * W; double* ts = tc + W;      // xr[x][RB]
  const int64_t row0 = (int64_t)blockIdx.x * DFT64_RB;
  for (int i = threadIdx.x; i < DFT64_RB * W; i += 256) {
    const int r = i / W, x = i - r * W;
    xr[x * DFT64_RB + r] = row0 + r < nrows ? (double)in[(row0 + r) * W + x] : 0.0;
  }
  for (int x = threadIdx.x; x < W; x += 256) { tc[x] = cs[x]; ts[x] = sn[x]; }
  __syncthreads();
  for (int v = threadIdx.x; v < W; v += 256) {
    double ar[DFT64_RB], ai[DFT64_RB];
#pragma unroll
    for (int r = 0; r < DFT64_RB; ++r) { ar[r] = 0.0; ai[r] = 0.0; }
    int idx = 0;
    for (int x = 0; x < W; ++x) {
      const double c = tc[idx], sv = ts[idx];          // e^{-2 pi i v x / W} = c - i sv
#pragma unroll
      for (int r = 0; r < DFT64_RB; ++r) {
        const double a = xr[x * DFT64_RB + r];
        ar[r] = fma(a, c, ar[r]);
        ai[r] = fma(-a, sv, ai[r]);
      }
      idx += v; if (idx >= W) idx -= W;
    }
#pragma unroll
    for (int r = 0; r < DFT64_RB; ++r)
      if (row0 + r < nrows) { out_re[(row0 + r) * W + v] = ar[r]; out_im[(row0 + r) * W + v] = ai[r]; }
  }
}
// columns fused with the normalisation to the unit phase: workgroup = (plane, 64-wide x strip, 4 x 8 output rows); lane =
// x (coalesced), every wave owns 8 output rows u whose twiddles are wave-uniform LDS reads
constexpr int DFT64_UB = 8;
__global__ __launch_bounds__(256) void dft_cols_unit_f64_k(const double* __restrict__ in_re, const double* __restrict__ in_im,
                                                           float* __restrict__ ure, float* __restrict__ uim, int H, int W,
                                                           const double* __restrict__ cs, const double* __restrict__ sn) {
  extern __shared__ __attribute__((aligned(16))) double sh64[];
  double* tc = sh64; double* ts = sh64 + H;
  for (int y = threadIdx.x; y < H; y += 256) { tc[y] = cs[y]; ts[y] = -sn[y]; }
  __syncthreads();
  const int p = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int u0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * DFT64_UB;
  if (x >= W || u0 >= H) return;
  const double* br = in_re + (size_t)p * H * W + x;
  const double* bi = in_im + (size_t)p * H * W + x;
  double ar[DFT64_UB], ai[DFT64_UB];
  int idx[DFT64_UB], inc[DFT64_UB];
#pragma unroll
  for (int j = 0; j < DFT64_UB; ++j) { ar[j] = 0.0; ai[j] = 0.0; idx[j] = 0; inc[j] = u0 + j < H ? u0 + j : 0; }
  for (int y = 0; y < H; ++y) {
    const double vr = br[(size_t)y * W], vi = bi[(size_t)y * W];
#pragma unroll
    for (int j = 0; j < DFT64_UB; ++j) {
      const double c = tc[idx[j]], sv = ts[idx[j]];
      ar[j] += vr * c - vi * sv;
      ai[j] += vr * sv + vi * c;
      idx[j] += inc[j]; if (idx[j] >= H) idx[j] -= H;
    }
  }
#pragma unroll
  for (int j = 0; j < DFT64_UB; ++j) {
    const int u = u0 + j;
    if (u >= H) break;
    const size_t o = ((size_t)p * H + u) * W + x;
    const double mag = sqrt(ar[j] * ar[j] + ai[j] * ai[j]);
    if (mag > 0.0) { ure[o] = (float)(ar[j] / mag); uim[o] = (float)(ai[j] / mag); }
    else { ure[o] = 1.f; uim[o] = 0.f; }    // angle(0) = atan2(0,0) = 0
  }
}

__global__ void unit_phase_k(float* re, float* im, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float r = re[i], m = im[i];
    float mag = sqrtf(r * r + m * m);
    if (mag > 0.f) { re[i] = r / mag; im[i] = m / mag; }
    else { re[i] = 1.f; im[i] = 0.f; }    // angle(0) = atan2(0,0) = 0
  }
}

// A = |X| * U   (in place into tmp)
__global__ void mag_times_phase_k(const float* xre, const float* xim, const float* ure, const float* uim,
                                  float* are, float* aim, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float mag = sqrtf(xre[i] * xre[i] + xim[i] * xim[i]);
    are[i] = mag * ure[i];
    aim[i] = mag * uim[i];
  }
}

__global__ void blend_k(const float* x, const float* adj, float* out, float alpha, float beta, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = __fadd_rn(__fmul_rn(alpha, x[i]), __fmul_rn(beta, adj[i]));
}

__global__ void axpby_k(const float* x, const float* e, float* out, float c1, float c2, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = __fmul_rn(c1, __fsub_rn(x[i], __fmul_rn(c2, e[i])));
}

inline unsigned grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)g;
}

}  // namespace

void launch_quantize_u8(const float* x, uint8_t* u8, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(quantize_u8_k, dim3(grid_for(n)), dim3(256), 0, s, x, u8, n);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_ddrm_update(const float* x_theta, const uint8_t* dec_u8, const float* y, const float* noise, float* x_out,
                        int64_t n_per_image, int B, float eta, float eta_b, float one_minus_eta_b, float noise_coeff,
                        int step_i, int steps, int last, hipStream_t s) {
  // noise_scale = t.float() * coeff with t = float(i)/steps  (webp_inference.py:568,588): two fp32 roundings
  float t = (float)step_i / (float)steps;
  float noise_scale = t * noise_coeff;
  int64_t n = n_per_image * B;
  hipLaunchKernelGGL(ddrm_update_k, dim3(grid_for(n)), dim3(256), 0, s, x_theta, dec_u8, y, noise, x_out, n, eta,
                     eta_b, one_minus_eta_b, noise_scale, last);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_fill_t(float* t, int B, int step_i, int steps, hipStream_t s) {
  float v = (float)step_i / (float)steps;
  hipLaunchKernelGGL(fill_t_k, dim3(cdiv(B, 64)), dim3(64), 0, s, t, B, v);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
  hipLaunchKernelGGL(randn_k, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, out, n, seed, offset, (const StepParams*)nullptr);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_step_params_set(StepParams* p, int i, int steps, float coeff, uint64_t seed_base, uint64_t rng_offset,
                            const float* noise, int64_t noise_stride, hipStream_t s) {
  hipLaunchKernelGGL(step_params_set_k, dim3(1), dim3(1), 0, s, p, i, steps, coeff, (unsigned long long)seed_base,
                     (unsigned long long)rng_offset, noise, (long long)noise_stride);
  DDPMIR_HIP_CHECK(hipGetLastError());
}
void launch_step_params_advance(StepParams* p, hipStream_t s) {
  hipLaunchKernelGGL(step_params_advance_k, dim3(1), dim3(1), 0, s, p);
  DDPMIR_HIP_CHECK(hipGetLastError());
}
void launch_fill_t_p(float* t, int B, const StepParams* p, hipStream_t s) {
  hipLaunchKernelGGL(fill_t_p_k, dim3(cdiv(B, 64)), dim3(64), 0, s, t, B, p);
  DDPMIR_HIP_CHECK(hipGetLastError());
}
void launch_randn_p(float* out, int64_t n, const StepParams* p, hipStream_t s) {
  hipLaunchKernelGGL(randn_k, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, out, n, (uint64_t)0, (uint64_t)0, p);
  DDPMIR_HIP_CHECK(hipGetLastError());
}
void launch_ddrm_update_p(const float* x_theta, const uint8_t* dec_u8, const float* y, const float* dev_noise, int64_t lane_off,
                          float* x_out, int64_t n, float eta, float eta_b, float one_minus_eta_b, const StepParams* p,
                          hipStream_t s) {
  hipLaunchKernelGGL(ddrm_update_p_k, dim3(grid_for(n)), dim3(256), 0, s, x_theta, dec_u8, y, dev_noise, (long long)lane_off,
                     x_out, n, eta, eta_b, one_minus_eta_b, p);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

// rows wider than ~900 pixels need more than the default 64 KiB of dynamic LDS
static void dft_rows_lds(int W) {
  const size_t bytes = (size_t)(2 * DFT_RB + 2) * W * sizeof(float);
  if (bytes > 160 * 1024) throw std::invalid_argument("phase_consistency: image width too large for the DFT row kernel");
  if (bytes > 64 * 1024) {   // per launch: the limit depends on W (rare: rows wider than ~900 pixels)
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dft_rows_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)bytes));
  }
}

void launch_dft2_forward(const float* x, float* re, float* im, float* tmp_re, float* tmp_im, int P, const DftPlan& d,
                         hipStream_t s) {
  const int H = d.H, W = d.W;
  const int64_t nrows = (int64_t)P * H;
  dft_rows_lds(W);
  hipLaunchKernelGGL(dft_rows_k, dim3((unsigned)cdiv64(nrows, DFT_RB)), dim3(256), (2 * DFT_RB + 2) * W * sizeof(float), s,
                     x, (const float*)nullptr, tmp_re, tmp_im, W, nrows, d.cos_w, d.sin_w, -1.0f, 1.0f);
  hipLaunchKernelGGL(dft_cols_k, dim3(cdiv(W, 64), cdiv(H, 4 * DFT_UB), P), dim3(256), 2 * H * sizeof(float), s, tmp_re,
                     tmp_im, re, im, H, W, d.cos_h, d.sin_h, -1.0f, 1.0f, 0);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_unit_phase(float* re, float* im, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(unit_phase_k, dim3(grid_for(n)), dim3(256), 0, s, re, im, n);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_phase_apply_inverse(const float* x, const float* xre, const float* xim, const float* ure,
                                const float* uim, float* tmp_re, float* tmp_im, float* tmp2_re, float* tmp2_im,
                                float* out, int P, const DftPlan& d, float alpha, float one_minus_alpha, hipStream_t s) {
  const int H = d.H, W = d.W;
  const int64_t n = (int64_t)P * H * W;
  hipLaunchKernelGGL(mag_times_phase_k, dim3(grid_for(n)), dim3(256), 0, s, xre, xim, ure, uim, tmp_re, tmp_im, n);
  // inverse along columns (u -> y), then rows (v -> x); 1/(H*W) split over the two passes
  hipLaunchKernelGGL(dft_cols_k, dim3(cdiv(W, 64), cdiv(H, 4 * DFT_UB), P), dim3(256), 2 * H * sizeof(float), s, tmp_re,
                     tmp_im, tmp2_re, tmp2_im, H, W, d.cos_h, d.sin_h, 1.0f, 1.0f / (float)H, 0);
  const int64_t nrows = (int64_t)P * H;
  dft_rows_lds(W);
  hipLaunchKernelGGL(dft_rows_k, dim3((unsigned)cdiv64(nrows, DFT_RB)), dim3(256), (2 * DFT_RB + 2) * W * sizeof(float), s,
                     tmp2_re, tmp2_im, tmp_re, tmp_im, W, nrows, d.cos_w, d.sin_w, 1.0f, 1.0f / (float)W);
  // python: alpha * x + (1 - alpha) * adjusted ; (1 - alpha) is evaluated in double by the caller
  hipLaunchKernelGGL(blend_k, dim3(grid_for(n)), dim3(256), 0, s, x, tmp_re, out, alpha, one_minus_alpha, n);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_ref_unit_phase_f64(const float* y, float* ure, float* uim, double* tmp_re, double* tmp_im, int P, const DftPlan64& d,
                               hipStream_t s) {
  const int64_t nrows = (int64_t)P * d.H;
  const size_t lds_r = (size_t)(DFT64_RB + 2) * d.W * sizeof(double), lds_c = (size_t)2 * d.H * sizeof(double);
  // 160 KiB of LDS per workgroup: rows up to 3 413 pixels wide, columns up to 10 240 high (the fp32 transforms of the phase fix
  // itself stop at 160 KiB too: dft_rows_lds).  Past the default 64 KiB the limit is raised per launch, as it depends on W / H.
  if (lds_r > 160 * 1024 || lds_c > 160 * 1024) throw std::invalid_argument("phase_consistency: image too large for the fp64 reference-phase kernels (width <= 3413, height <= 10240)");
  if (lds_r > 64 * 1024)
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dft_rows_f64_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
  if (lds_c > 64 * 1024)
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dft_cols_unit_f64_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
  hipLaunchKernelGGL(dft_rows_f64_k, dim3((unsigned)cdiv64(nrows, DFT64_RB)), dim3(256), lds_r, s, y, tmp_re, tmp_im, d.W, nrows, d.cos_w,
                     d.sin_w);
  hipLaunchKernelGGL(dft_cols_unit_f64_k, dim3(cdiv(d.W, 64), cdiv(d.H, 4 * DFT64_UB), P), dim3(256), lds_c, s, (const double*)tmp_re,
                     (const double*)tmp_im, ure, uim, d.H, d.W, d.cos_h, d.sin_h);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_axpby(const float* x, const float* e, float* out, float c1, float c2, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(axpby_k, dim3(grid_for(n)), dim3(256), 0, s, x, e, out, c1, c2, n);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

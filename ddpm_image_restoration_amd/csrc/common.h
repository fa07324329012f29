// Common definitions for the gfx950 (MI355X) restoration sampler kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace ddpmir {

// ---- activation codes shared by prologues / epilogues
enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_LRELU = 3, ACT_RELU = 4, ACT_SIGMOID = 5, ACT_TANH = 6 };
// ---- static in-block frequency masks (webp_inference.py:241-252)
enum MaskMode : int { MASK_NONE = 0, MASK_LOW = 1, MASK_HIGH = 2 };
// ---- output store layouts
enum StoreMode : int { STORE_NHWC = 0, STORE_NCHW = 1 };

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_SILU: return v / (1.0f + __expf(-v));
    case ACT_LRELU: return fmaxf(v, 0.2f * v);            // = v > 0 ? v : 0.2 v for every input (slope < 1), one compare-select less
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// GELU for a result that is rounded to bf16: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, three orders below the bf16
// rounding step) -- one v_rcp_f32, one v_exp_f32 and seven FMAs instead of ocml's erff (~45 instructions).  The norm2-apply pass
// over a full-resolution tensor was bound by exactly that arithmetic (134 M elements: 156 us for 537 MB, 3.4 TB/s).
__device__ __forceinline__ float gelu_bf16_grade(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = 1.0f - poly * __expf(-x * x);
  return 0.5f * v * (1.0f + copysignf(e, v));
}
// Sigmoid / SiLU of the bf16 pipeline: 1 / (1 + e^-v) through v_rcp_f32 (1 ulp) instead of the IEEE division, which expands to ten
// instructions per value -- a fifth of the fused frequency-gate kernel, which is bound by instruction issue.
__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// activations of the bf16 / fp8 modes (every kernel of those modes calls this one, so fused and unfused paths agree bit for bit)
__device__ __forceinline__ float act16(float v, int act) {
  switch (act) {
    case ACT_GELU: return gelu_bf16_grade(v);
    case ACT_SILU: return v * sigmoid_fast(v);
    case ACT_SIGMOID: return sigmoid_fast(v);
    default: return act_apply(v, act);
  }
}


// low-frequency membership of pixel (y,x) for block size bs and corner size `low`, including the
// partial-block rule low_size = max(1, min(low, i_end-i, j_end-j))  (webp_inference.py:247)
__device__ __forceinline__ bool is_low_freq(int y, int x, int H, int W, int bs, int low) {
  int iy = y % bs, ix = x % bs;
  int by = y - iy, bx = x - ix;
  int eh = min(bs, H - by), ew = min(bs, W - bx);
  int ls = max(1, min(low, min(eh, ew)));
  return iy < ls && ix < ls;
}

// four fp32 -> four OCP e4m3 bytes (round to nearest even, clamped to the finite range +-448)
__device__ __forceinline__ unsigned pack_e4m3x4(float4 v) {
  const float lim = 448.f;
  v.x = fminf(fmaxf(v.x, -lim), lim); v.y = fminf(fmaxf(v.y, -lim), lim);
  v.z = fminf(fmaxf(v.z, -lim), lim); v.w = fminf(fmaxf(v.w, -lim), lim);
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(v.x, v.y, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(v.z, v.w, r, true);
  return (unsigned)r;
}

// Division of a non-negative 32-bit index by a launch-time constant (H*W, W): the compiler expands a runtime-divisor "/" into
// ~40 vector instructions, and the GEMM kernels need several per thread in their prologue and per epilogue pass.  Round-up
// magic number (Granlund-Montgomery / Hacker's Delight 10-9): q = (umulhi(n, m) + n) >> s evaluated without overflow; exact for
// every n < 2^31.  Built on the host, carried in the kernel arguments.
struct FastDiv {
  unsigned d = 1, m = 0, s = 0;
  FastDiv() = default;
  explicit FastDiv(unsigned div) : d(div ? div : 1) {
    s = 0;
    while ((1ull << s) < d) ++s;                                  // s = ceil(log2 d)
    m = (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1);   // < 2^32
  }
  __host__ __device__ __forceinline__ unsigned div(unsigned n) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned t = __umulhi(n, m);
#else
    const unsigned t = (unsigned)(((unsigned long long)n * m) >> 32);
#endif
    return s == 0 ? n : (t + ((n - t) >> 1)) >> (s - 1);
  }
};

struct Status {
  int code = 0;
  std::string msg;
};

#define DDPMIR_HIP_CHECK(expr)                                                                     \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      char _b[512];                                                                                \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      throw std::runtime_error(_b);                                                                \
    }                                                                                              \
  } while (0)

// Experiment switches (A/B environment variables of the measurement tools under tools/) exist only in builds made with
// -DDDPMIR_EXPERIMENTS (build.py --experiments -> libddpmir_exp.so); the shipped library carries the production path alone.
#ifdef DDPMIR_EXPERIMENTS
inline bool exp_switch(const char* name) { return getenv(name) != nullptr; }
inline int exp_value(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#else
constexpr bool exp_switch(const char*) { return false; }
constexpr int exp_value(const char*) { return 0; }
#endif

// Function attributes (e.g. the dynamic LDS limit) belong to a (kernel, device) pair: contexts on different GPUs of one
// process each need them, possibly from different threads.  One instance per kernel instantiation.
struct PerDeviceOnce {
  std::mutex mu;
  unsigned long long done = 0;
  template <typename F>
  void run(F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) throw std::runtime_error("hipGetDevice failed");
    std::lock_guard<std::mutex> g(mu);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done & bit)) { f(); done |= bit; }
  }
};

// float -> OCP e4m3fn byte (1-4-3, bias 7, max finite 448, no infinities), round to nearest even, saturating: the host-side
// twin of v_cvt_pk_fp8_f32 on gfx950, used to pack fp8 weights
static inline unsigned char f32_to_e4m3(float f) {
  unsigned int u;
  memcpy(&u, &f, 4);
  const unsigned char sign = (unsigned char)((u >> 24) & 0x80u);
  const float a = fabsf(f);
  if (a != a) return (unsigned char)(sign | 0x7F);
  if (a >= 448.f) return (unsigned char)(sign | 0x7E);
  int e;
  (void)frexpf(a, &e);                       // a = m * 2^e, m in [0.5, 1)  ->  a = 1.x * 2^(e-1)
  int E = e - 1;
  if (a == 0.f) return sign;
  if (E < -6) {                              // subnormal range: multiples of 2^-9
    const int r = (int)nearbyintf(a * 512.f);
    return (unsigned char)(sign | (r >= 8 ? 0x08 : r));
  }
  int r = (int)nearbyintf(a / ldexpf(1.f, E - 3));   // mantissa with the hidden bit, 8 .. 16
  if (r == 16) { r = 8; E += 1; }
  unsigned bits = (unsigned)(((E + 7) << 3) | (r - 8));
  if (bits > 0x7E) bits = 0x7E;
  return (unsigned char)(sign | bits);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

}  // namespace ddpmir

// Host-buffer single-operator entry points (include/ddpmir_ops.h) used by the per-operator parity tests.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ddpmir.h"
#include "../../include/ddpmir_ops.h"
#include "kernels.h"

using namespace ddpmir;

namespace {
thread_local std::string g_err;
thread_local int g_bf16 = 0;
thread_local int g_epi_direct = 1;   // IGemm::epi_direct_opt of the GEMM ops below
thread_local int g_tile256 = -1;   // IGemm::tile256 of the GEMM ops below (-1 auto, 0 never, 1 wherever supported)
thread_local int g_sel = 0;

struct Dev {
  std::vector<void*> ptrs;
  ~Dev() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T>
  T* up(const T* host, size_t n) {
    if (!host) return nullptr;
    T* d = alloc<T>(n);
    DDPMIR_HIP_CHECK(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    return d;
  }
  template <typename T>
  T* alloc(size_t n) {
    void* d = nullptr;
    DDPMIR_HIP_CHECK(hipMalloc(&d, (n ? n : 1) * sizeof(T)));
    DDPMIR_HIP_CHECK(hipMemset(d, 0, (n ? n : 1) * sizeof(T)));
    ptrs.push_back(d);
    return static_cast<T*>(d);
  }
  template <typename T>
  void down(T* host, const T* dev, size_t n) {
    DDPMIR_HIP_CHECK(hipDeviceSynchronize());
    DDPMIR_HIP_CHECK(hipMemcpy(host, dev, n * sizeof(T), hipMemcpyDeviceToHost));
  }
};

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return DDPMIR_EINVAL;
  }
}
}  // namespace

extern "C" {

const char* ddpmir_ops_last_error(void) { return g_err.c_str(); }
void ddpmir_op_set_bf16(int enable) { g_bf16 = enable; }
void ddpmir_op_set_gate_select(int C) { g_sel = C; }
void ddpmir_op_set_tile256(int mode) { g_tile256 = mode; }
void ddpmir_op_set_epilogue_direct(int on) { g_epi_direct = on; }
long long ddpmir_op_gemm256_launches(void) { return gemm256_launch_count(); }

int ddpmir_op_conv(const float* x, int B, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout,
                   int ksize, const float* scale, const float* shift, int pro_act, int pro_mask, int bs, int low,
                   const float* tb, int epi_act, const float* mul, const float* rowscale, int epi_mask,
                   const float* res, int nchw_out, float* out) {
  return guarded([&] {
    Dev d;
    const int cinp = round_up(Cin, 4), taps = ksize * ksize;
    const size_t M = (size_t)B * H * W;
    // pad channels to a multiple of 4 exactly like the model's weight packer
    std::vector<float> xp(M * cinp, 0.f), wp((size_t)taps * Cout * cinp, 0.f), sp, hp;
    for (size_t m = 0; m < M; ++m)
      for (int c = 0; c < Cin; ++c) xp[m * cinp + c] = x[m * Cin + c];
    for (int o = 0; o < Cout; ++o)
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < taps; ++t) wp[((size_t)t * Cout + o) * cinp + c] = w_oihw[((size_t)o * Cin + c) * taps + t];
    if (g_bf16 == 3) {
      // fp8 variant of the DMA GEMM: x -> e4m3 by the operand pass, weights -> e4m3 with a per-tensor power-of-two scale
      if (pro_mask != MASK_NONE || nchw_out || (Cout % 4) != 0 || (Cin % 16) != 0 || Cout <= 32) throw std::runtime_error("fp8 gemm op: unsupported shape");
      float amax = 0.f;
      for (float v : wp) amax = std::max(amax, std::fabs(v));
      const int kexp = amax > 0.f ? (int)std::floor(std::log2(448.0 / (double)amax)) : 0;
      std::vector<unsigned char> w8((size_t)taps * Cout * Cin, 0);
      for (size_t i = 0; i < w8.size(); ++i) w8[i] = f32_to_e4m3(wp[i] * std::ldexp(1.0f, kexp));     // cinp == Cin here
      float* dx = d.up(xp.data(), xp.size());
      float *dsc = nullptr, *dsh = nullptr;
      if (scale) { dsc = d.up(scale, (size_t)B * Cin); dsh = d.up(shift, (size_t)B * Cin); }
      unsigned char* a8 = d.alloc<unsigned char>(M * Cin + 16);
      launch_gn_apply16(dx, Cin, B, H * W, Cin, dsc, dsh, pro_act, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, a8, Cin);
      IGemm p;
      p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.taps = taps; p.bias = d.up(bias, Cout);
      p.A16 = a8; p.lda16 = Cin; p.Wt16 = d.up(w8.data(), w8.size()); p.f8 = 1; p.acc_scale = std::ldexp(1.0f, -kexp);
      p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
      p.tb = d.up(tb, (size_t)B * Cout); p.ldtb = Cout; p.epi_act = epi_act; p.bs = bs; p.low = low;
      p.mul = d.up(mul, M * Cout); p.ldmul = Cout; p.rowscale = d.up(rowscale, B); p.epi_mask = epi_mask;
      p.res = d.up(res, M * Cout); p.ldres = Cout;
      float* o = d.alloc<float>(M * Cout);
      p.out = o; p.ldo = Cout; p.store = STORE_NHWC;
      p.tile256 = g_tile256; p.epi_direct_opt = g_epi_direct; launch_gemm16(p, nullptr);
      d.down(out, o, M * Cout);
      return;
    }
    if (g_bf16 == 2) {
      // DMA-streamed bf16 GEMM (gemm16.hip): operands pre-converted to bf16 by gn_apply16, channels padded to 8
      if (pro_mask != MASK_NONE || nchw_out || (Cout % 4) != 0) throw std::runtime_error("gemm16 op: unsupported options");
      const int cin8 = round_up(Cin, 8);
      std::vector<unsigned short> w8((size_t)taps * Cout * cin8, 0);
      for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < Cin; ++c)
          for (int t = 0; t < taps; ++t) {
            unsigned int u;
            memcpy(&u, &w_oihw[((size_t)o * Cin + c) * taps + t], 4);
            w8[((size_t)t * Cout + o) * cin8 + c] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
          }
      float* dx = d.up(xp.data(), xp.size());
      float *dsc = nullptr, *dsh = nullptr;
      if (scale) {
        sp.assign((size_t)B * cinp, 0.f); hp.assign((size_t)B * cinp, 0.f);
        for (int b = 0; b < B; ++b)
          for (int c = 0; c < Cin; ++c) { sp[b * cinp + c] = scale[b * Cin + c]; hp[b * cinp + c] = shift[b * Cin + c]; }
        dsc = d.up(sp.data(), sp.size()); dsh = d.up(hp.data(), hp.size());
      }
      unsigned short* a16 = d.alloc<unsigned short>(M * cin8 + 8);
      launch_gn_apply16(dx, cinp, B, H * W, cinp, dsc, dsh, pro_act, a16, cin8, nullptr, 0, nullptr);
      IGemm p;
      p.B = B; p.H = H; p.W = W; p.Cin = cin8; p.Cout = Cout; p.taps = taps; p.bias = d.up(bias, Cout);
      p.A16 = a16; p.lda16 = cin8; p.Wt16 = d.up(w8.data(), w8.size()); p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
      p.tb = d.up(tb, (size_t)B * Cout); p.ldtb = Cout; p.epi_act = epi_act; p.bs = bs; p.low = low;
      // gate-select mode: mul / res / out have Cout/2 columns (see IGemm::sel_C)
      const int Co = g_sel ? g_sel : Cout;
      p.sel_C = g_sel;
      p.mul = d.up(mul, M * Co); p.ldmul = Co; p.rowscale = d.up(rowscale, B); p.epi_mask = epi_mask;
      p.res = d.up(res, M * Co); p.ldres = Co;
      float* o = d.alloc<float>(M * Co);
      p.out = o; p.ldo = Co; p.store = STORE_NHWC;
      p.tile256 = g_tile256; p.epi_direct_opt = g_epi_direct; launch_gemm16(p, nullptr);
      d.down(out, o, M * Co);
      return;
    }
    IGemm p;
    p.A = d.up(xp.data(), xp.size()); p.lda = cinp; p.B = B; p.H = H; p.W = W; p.Cin = cinp;
    p.Wt = d.up(wp.data(), wp.size()); p.Cout = Cout; p.taps = taps; p.bias = d.up(bias, Cout);
    std::vector<unsigned short> w16;
    if (g_bf16) {
      w16.resize(wp.size());
      for (size_t i = 0; i < wp.size(); ++i) {
        unsigned int u;
        memcpy(&u, &wp[i], 4);
        w16[i] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
      }
      p.Wt16 = d.up(w16.data(), w16.size());
    }
    if (scale) {
      sp.assign((size_t)B * cinp, 0.f); hp.assign((size_t)B * cinp, 0.f);
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cin; ++c) { sp[b * cinp + c] = scale[b * Cin + c]; hp[b * cinp + c] = shift[b * Cin + c]; }
      p.scale = d.up(sp.data(), sp.size()); p.shift = d.up(hp.data(), hp.size());
    }
    p.pro_act = pro_act; p.pro_mask = pro_mask; p.bs = bs; p.low = low;
    p.tb = d.up(tb, (size_t)B * Cout); p.ldtb = Cout; p.epi_act = epi_act;
    p.mul = d.up(mul, M * Cout); p.ldmul = Cout; p.rowscale = d.up(rowscale, B); p.epi_mask = epi_mask;
    p.res = d.up(res, M * Cout); p.ldres = Cout;
    float* o = d.alloc<float>(M * Cout);
    p.out = o; p.ldo = Cout; p.store = nchw_out ? STORE_NCHW : STORE_NHWC;
    launch_igemm(p, nullptr);
    d.down(out, o, M * Cout);
  });
}

int ddpmir_op_conv_time(int B, int H, int W, int Cin, int Cout, int ksize, int bf16, int gelu_norm, int residual,
                        int dbg, int iters, float* ms_per_call) {
  return guarded([&] {
    Dev d;
    const int taps = ksize * ksize;
    const size_t M = (size_t)B * H * W;
    float* x = d.alloc<float>(M * Cin + 4);
    launch_randn(x, (int64_t)(M * Cin), 11, 0, nullptr);
    float* w = d.alloc<float>((size_t)taps * Cout * Cin + 4);
    launch_randn(w, (int64_t)taps * Cout * Cin, 12, 0, nullptr);
    std::vector<float> wh((size_t)taps * Cout * Cin);
    d.down(wh.data(), w, wh.size());
    std::vector<unsigned short> w16(wh.size());
    for (size_t i = 0; i < wh.size(); ++i) {
      unsigned int u;
      memcpy(&u, &wh[i], 4);
      w16[i] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
    IGemm p;
    p.A = x; p.lda = Cin; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Wt = w; p.Cout = Cout; p.taps = taps;
    if (bf16) p.Wt16 = d.up(w16.data(), w16.size());
    if (bf16 == 2) {   // gemm16 path: bf16 A tensor made once outside the timed region; fp32 + bf16 outputs
      if (Cin % 8) throw std::runtime_error("gemm16 timing needs Cin % 8 == 0");
      unsigned short* a16 = d.alloc<unsigned short>(M * Cin + 8);
      launch_gn_apply16(x, Cin, B, H * W, Cin, nullptr, nullptr, ACT_NONE, a16, Cin, nullptr, 0, nullptr);
      p.A16 = a16; p.lda16 = Cin; p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
      if (gelu_norm) { p.out16 = d.alloc<unsigned short>(M * Cout + 8); p.ldo16 = Cout; }
      gelu_norm = 0;
    }
    if (bf16 == 3) {   // fp8 variant of the gemm16 path: e4m3 A tensor and weights made once outside the timed region
      if (Cin % 16 || Cout <= 32) throw std::runtime_error("fp8 gemm timing needs Cin % 16 == 0 and Cout > 32");
      unsigned char* a8 = d.alloc<unsigned char>(M * Cin + 16);
      launch_gn_apply16(x, Cin, B, H * W, Cin, nullptr, nullptr, ACT_NONE, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, a8, Cin);
      std::vector<unsigned char> w8(wh.size());
      for (size_t i = 0; i < wh.size(); ++i) w8[i] = f32_to_e4m3(wh[i] * 64.f);
      p.A16 = a8; p.lda16 = Cin; p.Wt16 = d.up(w8.data(), w8.size()); p.f8 = 1; p.acc_scale = 1.f / 64.f;
      p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
      if (gelu_norm) { p.out8 = d.alloc<unsigned char>(M * Cout + 16); p.ldo8 = Cout; }
      gelu_norm = 0;
    }
    std::vector<float> ones((size_t)B * Cin, 1.0f), zeros((size_t)B * Cin, 0.1f), bias(Cout, 0.01f);
    p.bias = d.up(bias.data(), bias.size());
    if (gelu_norm) { p.scale = d.up(ones.data(), ones.size()); p.shift = d.up(zeros.data(), zeros.size()); p.pro_act = ACT_GELU; }
    float* o = d.alloc<float>(M * Cout);
    if (residual) { p.res = o; p.ldres = Cout; }
    p.out = o; p.ldo = Cout; p.dbg = dbg & 127;
    if ((dbg & 128) && (p.out16 || p.out8)) { p.out = nullptr; p.res = nullptr; }   // bit 7: the bf16 / e4m3 copy is the only output (conv1 / conv2 of a block)
    hipEvent_t a, b;
    DDPMIR_HIP_CHECK(hipEventCreate(&a));
    DDPMIR_HIP_CHECK(hipEventCreate(&b));
    p.tile256 = g_tile256; p.epi_direct_opt = g_epi_direct;
    auto run = [&] { if (bf16 >= 2) launch_gemm16(p, nullptr); else launch_igemm(p, nullptr); };
    run();
    DDPMIR_HIP_CHECK(hipDeviceSynchronize());
    DDPMIR_HIP_CHECK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) run();
    DDPMIR_HIP_CHECK(hipEventRecord(b, nullptr));
    DDPMIR_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    DDPMIR_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    *ms_per_call = ms / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  });
}

int ddpmir_op_conv_gnstats_bf16(const float* x, int B, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout,
                                int ksize, int groups, float* out, double* sums) {
  return guarded([&] {
    Dev d;
    const int taps = ksize * ksize, c8 = round_up(Cin, 8);
    const size_t M = (size_t)B * H * W;
    auto bf = [](float f) {
      unsigned int u;
      memcpy(&u, &f, 4);
      return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    };
    std::vector<unsigned short> wa((size_t)taps * Cout * c8, 0), xa(M * c8, 0);
    for (int o = 0; o < Cout; ++o)
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < taps; ++t) wa[((size_t)t * Cout + o) * c8 + c] = bf(w_oihw[((size_t)o * Cin + c) * taps + t]);
    for (size_t m = 0; m < M; ++m)
      for (int c = 0; c < Cin; ++c) xa[m * c8 + c] = bf(x[m * Cin + c]);
    IGemm p;
    p.B = B; p.H = H; p.W = W; p.Cin = c8; p.Cout = Cout; p.taps = taps; p.bias = d.up(bias, Cout);
    p.A16 = d.up(xa.data(), xa.size()); p.lda16 = c8; p.Wt16 = d.up(wa.data(), wa.size());
    p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
    float* o = d.alloc<float>(M * Cout);
    p.out = o; p.ldo = Cout;
    if (!gemm16_can_fuse_gn(p, groups)) throw std::runtime_error("conv_gnstats: shape cannot fuse the statistics");
    const size_t nw = (size_t)(H * W / 32);                    // one partial per (sample, 32-row wave tile, group)
    double* ds = d.alloc<double>(gn_sums_doubles(B, H * W, groups));
    p.gn_sums = ds; p.gn_groups = groups;
    p.tile256 = g_tile256; p.epi_direct_opt = g_epi_direct; launch_gemm16(p, nullptr);
    d.down(out, o, M * Cout);
    std::vector<double> slots(gn_sums_doubles(B, H * W, groups));
    d.down(slots.data(), ds, slots.size());
    for (int b = 0; b < B; ++b)            // fold the wave-tile partials of every (sample, group)
      for (int g = 0; g < groups * 2; ++g) {
        double t = 0.0;
        for (size_t sl = 0; sl < nw; ++sl) t += slots[((size_t)b * nw + sl) * groups * 2 + g];
        sums[(size_t)b * groups * 2 + g] = t;
      }
  });
}

int ddpmir_op_conv_tile(const float* x, int B, int H, int W, int Cin, int in16, const float* scale, const float* shift, int act,
                        const float* w_oihw, const float* bias, const float* tb, int Cout, int groups, int want16, float* out, float* out16f,
                        double* sums, float* raw16f) {
  return guarded([&] {
    Dev d;
    const size_t M = (size_t)B * H * W;
    auto bf = [](float f) {
      unsigned int u;
      memcpy(&u, &f, 4);
      return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    };
    auto unbf = [](unsigned short h) { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; };
    std::vector<unsigned short> wa((size_t)9 * Cout * Cin, 0);
    for (int o = 0; o < Cout; ++o)
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < 9; ++t) wa[((size_t)t * Cout + o) * Cin + c] = bf(w_oihw[((size_t)o * Cin + c) * 9 + t]);
    IGemm p;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.taps = 9; p.bias = d.up(bias, Cout);
    p.Wt16 = d.up(wa.data(), wa.size());
    if (tb) { p.tb = d.up(tb, (size_t)B * Cout); p.ldtb = Cout; }
    float* o32 = out ? d.alloc<float>(M * Cout) : nullptr;
    unsigned short* o16 = want16 ? d.alloc<unsigned short>(M * Cout + 8) : nullptr;
    p.out = o32; p.ldo = Cout; p.out16 = o16; p.ldo16 = Cout;
    double* ds = nullptr;
    if (groups > 0) { ds = d.alloc<double>(gn_sums_doubles(B, H * W, groups)); p.gn_sums = ds; p.gn_groups = groups; }
    const void* xin;
    if (in16) {
      std::vector<unsigned short> xa(M * Cin);
      for (size_t i = 0; i < xa.size(); ++i) xa[i] = bf(x[i]);
      xin = d.up(xa.data(), xa.size());
    } else {
      xin = d.up(x, M * Cin);
    }
    unsigned short* r16 = raw16f ? d.alloc<unsigned short>(M * Cin + 8) : nullptr;
    if (!conv_tile_supported(p)) throw std::runtime_error("conv_tile: unsupported shape or epilogue");
    launch_conv_tile(p, xin, Cin, in16 != 0, scale ? d.up(scale, (size_t)B * Cin) : nullptr, shift ? d.up(shift, (size_t)B * Cin) : nullptr, act, r16, Cin, nullptr);
    if (out) d.down(out, o32, M * Cout);
    if (want16) {
      std::vector<unsigned short> h(M * Cout);
      d.down(h.data(), o16, h.size());
      for (size_t i = 0; i < h.size(); ++i) out16f[i] = unbf(h[i]);
    }
    if (raw16f) {
      std::vector<unsigned short> h(M * Cin);
      d.down(h.data(), r16, h.size());
      for (size_t i = 0; i < h.size(); ++i) raw16f[i] = unbf(h[i]);
    }
    if (groups > 0) {
      const size_t nw = (size_t)(H * W / 32);
      std::vector<double> slots(gn_sums_doubles(B, H * W, groups));
      d.down(slots.data(), ds, slots.size());
      for (int b = 0; b < B; ++b)
        for (int g = 0; g < groups * 2; ++g) {
          double t = 0.0;
          for (size_t sl = 0; sl < nw; ++sl) t += slots[((size_t)b * nw + sl) * groups * 2 + g];
          sums[(size_t)b * groups * 2 + g] = t;
        }
    }
  });
}

int ddpmir_op_conv_dual_bf16(const float* x1, int B, int H, int W, int C1, const float* w1_oihw, int ksize, const float* x2,
                             int C2, const float* w2_oi, const float* bias, int Cout, float* out) {
  return guarded([&] {
    if (Cout % 4) throw std::runtime_error("conv_dual: Cout must be a multiple of 4");
    Dev d;
    const int taps = ksize * ksize, c18 = round_up(C1, 8), c28 = round_up(C2, 8);
    const size_t M = (size_t)B * H * W;
    auto bf = [](float f) {
      unsigned int u;
      memcpy(&u, &f, 4);
      return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    };
    std::vector<unsigned short> wa((size_t)taps * Cout * c18, 0), wb((size_t)Cout * c28, 0), xa(M * c18, 0), xb(M * c28, 0);
    for (int o = 0; o < Cout; ++o) {
      for (int c = 0; c < C1; ++c)
        for (int t = 0; t < taps; ++t) wa[((size_t)t * Cout + o) * c18 + c] = bf(w1_oihw[((size_t)o * C1 + c) * taps + t]);
      for (int c = 0; c < C2; ++c) wb[(size_t)o * c28 + c] = bf(w2_oi[(size_t)o * C2 + c]);
    }
    for (size_t m = 0; m < M; ++m) {
      for (int c = 0; c < C1; ++c) xa[m * c18 + c] = bf(x1[m * C1 + c]);
      for (int c = 0; c < C2; ++c) xb[m * c28 + c] = bf(x2[m * C2 + c]);
    }
    IGemm p;
    p.B = B; p.H = H; p.W = W; p.Cin = c18; p.Cout = Cout; p.taps = taps; p.bias = d.up(bias, Cout);
    p.A16 = d.up(xa.data(), xa.size()); p.lda16 = c18; p.Wt16 = d.up(wa.data(), wa.size());
    p.A16b = d.up(xb.data(), xb.size()); p.lda16b = c28; p.Wt16b = d.up(wb.data(), wb.size()); p.Cinb = c28;
    p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
    float* o = d.alloc<float>(M * Cout);
    p.out = o; p.ldo = Cout;
    p.tile256 = g_tile256; p.epi_direct_opt = g_epi_direct; launch_gemm16(p, nullptr);
    d.down(out, o, M * Cout);
  });
}

int ddpmir_op_conv_dual_fp8(const float* x1, int B, int H, int W, int C1, const float* w1_oihw, const float* x2, int C2,
                            const float* w2_oi, const float* bias, int Cout, float* out) {
  return guarded([&] {
    if ((Cout % 4) || (C1 % 16) || (C2 % 8) || Cout <= 32) throw std::runtime_error("conv_dual_fp8: unsupported shape");
    Dev d;
    const size_t M = (size_t)B * H * W;
    auto bf = [](float f) {
      unsigned int u;
      memcpy(&u, &f, 4);
      return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    };
    float amax = 0.f;
    for (size_t i = 0; i < (size_t)Cout * C1 * 9; ++i) amax = std::max(amax, std::fabs(w1_oihw[i]));
    const int kexp = amax > 0.f ? (int)std::floor(std::log2(448.0 / (double)amax)) : 0;
    const float sw = std::ldexp(1.0f, kexp);
    std::vector<unsigned char> w8((size_t)9 * Cout * C1, 0), x8(M * C1, 0);
    std::vector<unsigned short> wb((size_t)Cout * C2, 0), xb(M * C2, 0);
    for (int o = 0; o < Cout; ++o) {
      for (int c = 0; c < C1; ++c)
        for (int t = 0; t < 9; ++t) w8[((size_t)t * Cout + o) * C1 + c] = f32_to_e4m3(w1_oihw[((size_t)o * C1 + c) * 9 + t] * sw);
      for (int c = 0; c < C2; ++c) wb[(size_t)o * C2 + c] = bf(w2_oi[(size_t)o * C2 + c] * sw);
    }
    for (size_t i = 0; i < M * C1; ++i) x8[i] = f32_to_e4m3(x1[i]);
    for (size_t i = 0; i < M * C2; ++i) xb[i] = bf(x2[i]);
    IGemm p;
    p.B = B; p.H = H; p.W = W; p.Cin = C1; p.Cout = Cout; p.taps = 9; p.bias = d.up(bias, Cout);
    p.A16 = d.up(x8.data(), x8.size()); p.lda16 = C1; p.Wt16 = d.up(w8.data(), w8.size()); p.f8 = 1; p.acc_scale = std::ldexp(1.0f, -kexp);
    p.A16b = d.up(xb.data(), xb.size()); p.lda16b = C2; p.Wt16b = d.up(wb.data(), wb.size()); p.Cinb = C2;
    p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
    float* o = d.alloc<float>(M * Cout);
    p.out = o; p.ldo = Cout;
    p.tile256 = g_tile256; p.epi_direct_opt = g_epi_direct; launch_gemm16(p, nullptr);
    d.down(out, o, M * Cout);
  });
}

int ddpmir_op_qkv_attention_bf16(const float* x, int B, int N, int C, int heads, const float* w_in, const float* b_in,
                                 int fused, float* out) {
  return guarded([&] {
    if (C % 8 || !attention_bf16_supported(N, C, heads)) throw std::runtime_error("qkv_attention: unsupported shape");
    if (fused && !attention_bf16_prepack_supported(N, C, heads)) throw std::runtime_error("qkv_attention: shape cannot be fused");
    Dev d;
    const size_t M = (size_t)B * N;
    std::vector<unsigned short> w16((size_t)3 * C * C);
    for (size_t i = 0; i < w16.size(); ++i) {
      unsigned int u;
      memcpy(&u, &w_in[i], 4);
      w16[i] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
    float* dx = d.up(x, M * C);
    unsigned short* a16 = d.alloc<unsigned short>(M * C + 8);
    launch_gn_apply16(dx, C, B, N, C, nullptr, nullptr, ACT_NONE, a16, C, nullptr, 0, nullptr);
    void* ws = d.alloc<unsigned char>(attention_bf16_workspace(B, N, C, heads));
    IGemm p;
    p.B = B; p.H = 1; p.W = N; p.Cin = C; p.Cout = 3 * C; p.taps = 1; p.bias = d.up(b_in, (size_t)3 * C);
    p.A16 = a16; p.lda16 = C; p.Wt16 = d.up(w16.data(), w16.size()); p.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); p.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
    float* qkv = nullptr;
    if (fused) {
      const AttnPack16 t = attention_bf16_pack_targets(B, N, C, heads, ws);
      p.pk_q = t.q; p.pk_kv = t.kv; p.pk_kmax = t.kmax; p.pk_nt = t.ntiles; p.pk_qscale = t.qscale; p.pk_D = t.D; p.pk_heads = t.heads; p.pk_qmax = t.qmax2;
    } else {
      qkv = d.alloc<float>(M * 3 * C);
      p.out = qkv; p.ldo = 3 * C;
    }
    p.tile256 = g_tile256; p.epi_direct_opt = g_epi_direct; launch_gemm16(p, nullptr);
    float* o = d.alloc<float>(M * C);
    launch_attention_bf16(qkv, o, C, B, N, C, heads, ws, nullptr);
    d.down(out, o, M * C);
  });
}

int ddpmir_op_gn_scale_shift(const float* x, int B, int HW, int C, int G, const float* gamma, const float* beta,
                             float* scale, float* shift) {
  return guarded([&] {
    Dev d;
    const int Cs = round_up(C, 4);
    std::vector<float> xp((size_t)B * HW * Cs, 0.f);
    for (size_t m = 0; m < (size_t)B * HW; ++m)
      for (int c = 0; c < C; ++c) xp[m * Cs + c] = x[m * C + c];
    float* dx = d.up(xp.data(), xp.size());
    double* part = d.alloc<double>((size_t)B * 8 * gn_splits(HW, C) * 2 + 64);
    float* sc = d.alloc<float>((size_t)B * Cs);
    float* sh = d.alloc<float>((size_t)B * Cs);
    launch_gn_scale_shift(dx, Cs, B, HW, C, G, d.up(gamma, C), d.up(beta, C), part, sc, sh, nullptr);
    std::vector<float> s1((size_t)B * Cs), s2((size_t)B * Cs);
    d.down(s1.data(), sc, s1.size());
    d.down(s2.data(), sh, s2.size());
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c) { scale[b * C + c] = s1[b * Cs + c]; shift[b * C + c] = s2[b * Cs + c]; }
  });
}

int ddpmir_op_attention(const float* qkv, int B, int N, int C, int heads, float* out) {
  return guarded([&] {
    Dev d;
    float* q = d.up(qkv, (size_t)B * N * 3 * C);
    float* o = d.alloc<float>((size_t)B * N * C);
    launch_attention(q, o, C, B, N, C, heads, nullptr);
    d.down(out, o, (size_t)B * N * C);
  });
}

int ddpmir_op_attention_bf16(const float* qkv, int B, int N, int C, int heads, float* out) {
  return guarded([&] {
    if (!attention_bf16_supported(N, C, heads)) throw std::runtime_error("attention_bf16: unsupported shape");
    Dev d;
    float* q = d.up(qkv, (size_t)B * N * 3 * C);
    float* o = d.alloc<float>((size_t)B * N * C);
    unsigned char* ws = d.alloc<unsigned char>(attention_bf16_workspace(B, N, C, heads) + 256);
    launch_attention_bf16(q, o, C, B, N, C, heads, ws, nullptr);
    d.down(out, o, (size_t)B * N * C);
  });
}

int ddpmir_op_attention_bf16_ex(const float* qkv, int B, int N, int C, int heads, float nm_bound, float* out, int64_t* stats) {
  return guarded([&] {
    if (!attention_bf16_supported(N, C, heads)) throw std::runtime_error("attention_bf16: unsupported shape");
    Dev d;
    float* q = d.up(qkv, (size_t)B * N * 3 * C);
    float* o = d.alloc<float>((size_t)B * N * C);
    unsigned char* ws = d.alloc<unsigned char>(attention_bf16_workspace(B, N, C, heads) + 256);
    unsigned* taken = d.alloc<unsigned>(4);
    int64_t total = 0;
    AttnOpts ao;
    ao.nm_bound = nm_bound; ao.general_taken = taken; ao.general_total = &total;
    launch_attention_bf16(q, o, C, B, N, C, heads, ws, nullptr, nullptr, &ao);
    d.down(out, o, (size_t)B * N * C);
    unsigned t = 0;
    d.down(&t, taken, 1);
    if (stats) { stats[0] = (int64_t)t; stats[1] = total; }
  });
}

// Shader clock under load: one wave on a stream of its own reads s_memtime (shader cycles) against s_memrealtime (100 MHz) for a
// fixed wall time while the caller runs whatever it wants to characterise on other streams (tools/clock_under_load.py).
__global__ void clock_sampler_k(unsigned long long* out, unsigned long long wall_ticks) {
  const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  unsigned long long w = w0;
  while (w - w0 < wall_ticks) {                // bounded by the wall clock: the wave always exits
    __builtin_amdgcn_s_sleep(64);
    w = wall_clock64();
  }
  out[0] = __builtin_readcyclecounter() - c0;
  out[1] = w - w0;
}
static hipStream_t g_clk_stream = nullptr;
static unsigned long long* g_clk_buf = nullptr;
int ddpmir_op_clock_sample_begin(double ms) {
  return guarded([&] {
    if (ms <= 0 || ms > 5000) throw std::runtime_error("clock sample: 0 < ms <= 5000");
    if (!g_clk_stream) {
      DDPMIR_HIP_CHECK(hipStreamCreateWithFlags(&g_clk_stream, hipStreamNonBlocking));
      DDPMIR_HIP_CHECK(hipMalloc(&g_clk_buf, 16));
    }
    hipLaunchKernelGGL(clock_sampler_k, dim3(1), dim3(64), 0, g_clk_stream, g_clk_buf, (unsigned long long)(ms * 1e5));
    DDPMIR_HIP_CHECK(hipGetLastError());
  });
}
int ddpmir_op_clock_sample_end(double* ghz, double* ms) {
  return guarded([&] {
    if (!g_clk_stream) throw std::runtime_error("clock sample: begin was not called");
    DDPMIR_HIP_CHECK(hipStreamSynchronize(g_clk_stream));
    unsigned long long h[2];
    DDPMIR_HIP_CHECK(hipMemcpy(h, g_clk_buf, 16, hipMemcpyDeviceToHost));
    if (ghz) *ghz = (double)h[0] / ((double)h[1] * 10.0);
    if (ms) *ms = (double)h[1] * 1e-5;
  });
}

int ddpmir_op_attention_time(int B, int N, int C, int heads, int mode, int iters, float* ms_per_call) {
  return guarded([&] {
    Dev d;
    const size_t n = (size_t)B * N * 3 * C;
    float* q = d.alloc<float>(n + 4);
    launch_randn(q, (int64_t)n, 1234, 0, nullptr);
    float* o = d.alloc<float>((size_t)B * N * C);
    unsigned char* ws = mode ? d.alloc<unsigned char>(attention_bf16_workspace(B, N, C, heads) + 256) : nullptr;
    hipEvent_t a, b;
    DDPMIR_HIP_CHECK(hipEventCreate(&a));
    DDPMIR_HIP_CHECK(hipEventCreate(&b));
    auto run = [&] {
      if (mode) launch_attention_bf16(q, o, C, B, N, C, heads, ws, nullptr);
      else launch_attention(q, o, C, B, N, C, heads, nullptr);
    };
    run();
    DDPMIR_HIP_CHECK(hipDeviceSynchronize());
    DDPMIR_HIP_CHECK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) run();
    DDPMIR_HIP_CHECK(hipEventRecord(b, nullptr));
    DDPMIR_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    DDPMIR_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    *ms_per_call = ms / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  });
}

int ddpmir_op_block_transform(const float* x, int B, int H, int W, int C, int bs, const float* mats, int per_channel,
                              float mix_in, float mix_t, float* out) {
  return guarded([&] {
    Dev d;
    size_t n = (size_t)B * H * W * C;
    float* o = d.alloc<float>(n);
    launch_block_transform(d.up(x, n), C, o, C, B, H, W, C, bs, d.up(mats, (size_t)(per_channel ? C : 1) * bs * bs),
                           per_channel, mix_in, mix_t, nullptr);
    d.down(out, o, n);
  });
}

int ddpmir_op_freq_gates_bf16(const float* x, int B, int H, int W, int C, int bs, int low, const float* dct, const float* w1,
                              const float* b1, const float* w2, const float* b2, const float* boost, int fused, float* out,
                              const float* att, const float* wo, const float* bo) {
  return guarded([&] {
    if (C % 64) throw std::runtime_error("freq_gates: C must be a multiple of 64");
    Dev d;
    const size_t M = (size_t)B * H * W;
    auto bf = [](float f) {
      unsigned int u;
      memcpy(&u, &f, 4);
      return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    };
    std::vector<unsigned short> w1h((size_t)C * C), w2h((size_t)2 * C * C);
    for (size_t i = 0; i < w1h.size(); ++i) w1h[i] = bf(w1[i]);
    for (size_t i = 0; i < w2h.size(); ++i) w2h[i] = bf(w2[i]);
    const float* xd_in = d.up(x, M * C);
    const unsigned short* W1 = d.up(w1h.data(), w1h.size());
    const unsigned short* W2 = d.up(w2h.data(), w2h.size());
    const float *B1 = d.up(b1, C), *B2 = d.up(b2, 2 * C), *BO = d.up(boost, B), *T = d.up(dct, (size_t)bs * bs);
    unsigned short* S16 = d.alloc<unsigned short>(M * C);
    IGemm g0; g0.B = B; g0.H = H; g0.W = W; g0.bs = bs; g0.low = low; g0.taps = 1;
    g0.zero16 = d.alloc<unsigned char>(GEMM16_ZERO_BYTES); g0.splitk_ws = d.alloc<float>(GEMM16_SPLITK_WS_BYTES / 4); g0.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
    const unsigned short *ATT = nullptr, *WO = nullptr;
    const float* BOo = nullptr;
    if (att) {   // x = W_o att + b_o + x first (the attention's output projection and residual)
      std::vector<unsigned short> ah(M * C), wh((size_t)C * C);
      for (size_t i = 0; i < ah.size(); ++i) ah[i] = bf(att[i]);
      for (size_t i = 0; i < wh.size(); ++i) wh[i] = bf(wo[i]);
      ATT = d.up(ah.data(), ah.size()); WO = d.up(wh.data(), wh.size()); BOo = d.up(bo, C);
      if (!fused) {
        float* xo = d.alloc<float>(M * C);
        IGemm o = g0; o.A16 = ATT; o.lda16 = C; o.Wt16 = WO; o.bias = BOo; o.Cin = C; o.Cout = C; o.res = xd_in; o.ldres = C; o.out = xo; o.ldo = C;
        launch_gemm16(o, nullptr);
        xd_in = xo;
      }
    }
    if (fused) {
      launch_freq_gate_fused(xd_in, T, W1, B1, W2, B2, BO, S16, B, H, W, C, bs, low, nullptr, ATT, WO, BOo);
    } else {   // the kernels of Context::run_block16 before the fusion
      float* xd = d.alloc<float>(M * C);
      unsigned short* xdl16 = d.alloc<unsigned short>(M * C);
      unsigned short* hid16 = d.alloc<unsigned short>(M * C);
      launch_block_transform(xd_in, C, xd, C, B, H, W, C, bs, T, 0, 0.f, 1.f, nullptr, xdl16, nullptr, low);
      IGemm g = g0;
      IGemm a = g; a.A16 = xdl16; a.lda16 = C; a.Wt16 = W1; a.bias = B1; a.Cin = C; a.Cout = C; a.epi_act = ACT_LRELU; a.out16 = hid16; a.ldo16 = C;
      launch_gemm16(a, nullptr);
      IGemm e = g; e.A16 = hid16; e.lda16 = C; e.Wt16 = W2; e.bias = B2; e.Cin = C; e.Cout = 2 * C; e.epi_act = ACT_SIGMOID; e.mul = xd; e.ldmul = C;
      e.sel_C = C; e.rowscale = BO; e.res = xd_in; e.ldres = C; e.out16 = S16; e.ldo16 = C;
      launch_gemm16(e, nullptr);
    }
    std::vector<unsigned short> h(M * C);
    d.down(h.data(), S16, M * C);
    for (size_t i = 0; i < h.size(); ++i) { unsigned int u = (unsigned int)h[i] << 16; memcpy(&out[i], &u, 4); }
  });
}

// Inter-block stream passes (pointwise.hip stream_k).  mode 0: statistics of x alone (bf16 only); 1: 2x2 max-pool; 2: bilinear x2.
// f32 = 0: the bf16 activation stream -- x is rounded to bf16 on the way in, out (mode 1 / 2) is the bf16 result widened to
// fp32; f32 = 1: the fp32 stream.  scale / shift [B][C] are the GroupNorm affine folded from the partial sums the pass left
// (G groups over the C channels of the tensor the mode produces).
int ddpmir_op_stream16(int mode, const float* x, int B, int H, int W, int C, int ldx, int G, const float* gamma, const float* beta,
                       float* out, float* scale, float* shift, int f32) {
  return guarded([&] {
    Dev d;
    if (ldx < C) ldx = C;
    const size_t Mi = (size_t)B * H * W;
    const int Ho = mode == 1 ? H / 2 : mode == 2 ? 2 * H : H, Wo = mode == 1 ? W / 2 : mode == 2 ? 2 * W : W;
    const size_t Mo = (size_t)B * Ho * Wo;
    double* part = d.alloc<double>((size_t)B * 8 * 512 * 2 + 64);
    int splits = 0;
    unsigned short* o16 = nullptr;
    float* o32 = nullptr;
    if (f32) {
      if (mode == 0 || ldx != C) throw std::invalid_argument("fp32 stream: modes 1 and 2, dense rows");
      const float* dx = d.up(x, Mi * C);
      o32 = d.alloc<float>(Mo * C);
      if (mode == 1) splits = launch_maxpool2_stats(dx, C, o32, C, B, H, W, C, nullptr, G ? part : nullptr, G);
      else splits = launch_upsample2_stats(dx, C, o32, C, B, H, W, C, nullptr, G ? part : nullptr, G);
    } else {
      unsigned short* x16 = d.alloc<unsigned short>(Mi * ldx);
      launch_gn_apply16(d.up(x, Mi * C), C, B, H * W, C, nullptr, nullptr, ACT_NONE, x16, ldx, nullptr, 0, nullptr);
      o16 = d.alloc<unsigned short>(Mo * C);
      if (mode == 0) splits = launch_stats16(x16, ldx, B, H, W, C, G, G ? part : nullptr, nullptr);
      else if (mode == 1) splits = launch_maxpool2_16(x16, ldx, o16, C, B, H, W, C, nullptr, G ? part : nullptr, G);
      else splits = launch_upsample2_16(x16, ldx, o16, C, B, H, W, C, nullptr, G ? part : nullptr, G);
    }
    if (G) {
      float* sc = d.alloc<float>((size_t)B * C);
      float* sh = d.alloc<float>((size_t)B * C);
      launch_gn_from_partials(part, splits, B, (double)Ho * Wo * (C / G), C, G, d.up(gamma, C), d.up(beta, C), sc, sh, nullptr);
      d.down(scale, sc, (size_t)B * C);
      d.down(shift, sh, (size_t)B * C);
    }
    if (mode != 0 && f32) {
      d.down(out, o32, Mo * C);
    } else if (mode != 0) {
      std::vector<unsigned short> h(Mo * C);
      d.down(h.data(), o16, Mo * C);
      for (size_t i = 0; i < h.size(); ++i) { unsigned int u = (unsigned int)h[i] << 16; memcpy(&out[i], &u, 4); }
    }
  });
}

int ddpmir_op_maxpool2(const float* x, int B, int H, int W, int C, float* out) {
  return guarded([&] {
    Dev d;
    size_t no = (size_t)B * (H / 2) * (W / 2) * C;
    float* o = d.alloc<float>(no);
    launch_maxpool2(d.up(x, (size_t)B * H * W * C), C, o, C, B, H, W, C, nullptr);
    d.down(out, o, no);
  });
}

int ddpmir_op_upsample2(const float* x, int B, int H, int W, int C, float* out) {
  return guarded([&] {
    Dev d;
    size_t no = (size_t)B * H * W * C * 4;
    float* o = d.alloc<float>(no);
    launch_upsample2(d.up(x, (size_t)B * H * W * C), C, o, C, B, H, W, C, nullptr);
    d.down(out, o, no);
  });
}

int ddpmir_op_bilinear(const float* x, int B, int Hi, int Wi, int C, int Ho, int Wo, float* out) {
  return guarded([&] {
    Dev d;
    size_t no = (size_t)B * Ho * Wo * C;
    float* o = d.alloc<float>(no);
    launch_bilinear(d.up(x, (size_t)B * Hi * Wi * C), C, o, C, B, Hi, Wi, Ho, Wo, C, (float)Hi / (float)Ho,
                    (float)Wi / (float)Wo, 0, 1.0f, nullptr);
    d.down(out, o, no);
  });
}

int ddpmir_op_adaptive_avgpool(const float* x, int B, int H, int W, int C, int S, float* out) {
  return guarded([&] {
    Dev d;
    size_t no = (size_t)B * S * S * C;
    float* o = d.alloc<float>(no);
    float* dx = d.up(x, (size_t)B * H * W * C);
    if (pool8_supported(H, W, C) && (S == 1 || S == 2 || S == 4 || S == 8)) {   // the model's one-pass path
      float* sums = d.alloc<float>((size_t)B * 64 * C);
      launch_pool8_sums(dx, C, sums, B, H, W, C, nullptr);
      launch_pool_from8(sums, o, C, B, H, W, C, S, nullptr);
    } else {
      launch_adaptive_avgpool(dx, C, o, C, B, H, W, C, S, nullptr);
    }
    d.down(out, o, no);
  });
}

int ddpmir_op_time_embed(const float* t, int B, const float* w0, const float* b0, const float* w2, const float* b2,
                         float* temb) {
  return guarded([&] {
    Dev d;
    std::vector<float> wt(256 * 4, 0.f), bt(4, 0.f);
    TimeEmbedW w{d.up(w0, 1024 * 256), d.up(b0, 1024), d.up(w2, 256 * 1024), d.up(b2, 256), d.up(wt.data(), wt.size()),
                 d.up(bt.data(), 4), 4};
    float* hid = d.alloc<float>((size_t)B * 1024);
    float* te = d.alloc<float>((size_t)B * 256);
    float* tp = d.alloc<float>((size_t)B * 4);
    launch_time_embed(d.up(t, B), B, w, hid, te, tp, nullptr);
    d.down(temb, te, (size_t)B * 256);
  });
}

int ddpmir_op_quantize_u8(const float* x, int64_t n, uint8_t* out) {
  return guarded([&] {
    Dev d;
    uint8_t* o = d.alloc<uint8_t>(n);
    launch_quantize_u8(d.up(x, n), o, n, nullptr);
    d.down(out, o, n);
  });
}

int ddpmir_op_ddrm_update(const float* x_theta, const uint8_t* dec_u8, const float* y, const float* noise,
                          int64_t n_per_image, int B, double eta, double eta_b, double noise_coeff, int step_i, int steps,
                          float* out) {
  return guarded([&] {
    Dev d;
    size_t n = (size_t)n_per_image * B;
    float* o = d.alloc<float>(n);
    launch_ddrm_update(d.up(x_theta, n), d.up(dec_u8, n), d.up(y, n), d.up(noise, n), o, n_per_image, B, (float)eta,
                       (float)eta_b, (float)(1.0 - eta_b), (float)noise_coeff, step_i, steps, step_i == 0 ? 1 : 0, nullptr);
    d.down(out, o, n);
  });
}

int ddpmir_op_randn(int64_t n, uint64_t seed, uint64_t offset, float* out) {
  return guarded([&] {
    Dev d;
    float* o = d.alloc<float>(n + 4);
    launch_randn(o, n, seed, offset, nullptr);
    d.down(out, o, n);
  });
}

}  // extern "C"

// Host orchestration of the restoration U-Net forward and the DDRM sampling loop on one MI355X.
// Reference structure: webp_inference.py:303-327 (block), :369-399 (U-Net), :557-602 (sampler).
#include "model.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ddpmir {

namespace {

struct BlockDef { const char* name; int cin, cout, level; };
// webp_inference.py:337-356
const BlockDef kBlocks[13] = {
    {"down1", 3, 64, 0},          {"down2", 64, 128, 1},         {"down3", 128, 256, 2},
    {"down4", 256, 512, 3},       {"down5", 512, 512, 4},        {"bottleneck.0", 512, 1024, 5},
    {"bottleneck.1", 1024, 1024, 5}, {"bottleneck.2", 1024, 512, 5}, {"up1", 1024, 512, 4},
    {"up2", 1024, 256, 3},        {"up3", 512, 128, 2},          {"up4", 256, 64, 1},
    {"up5", 128, 64, 0}};

int gn_groups(int c) {  // webp_inference.py:277-279
  int g = std::min(8, c);
  while (c % g != 0 && g > 1) --g;
  return g;
}

ArchConsts consts_for(int arch) {
  switch (arch) {
    case DDPMIR_ARCH_WEBP: return {4, 4, 3, 0.15f, 1.9f, 0.1f, 0.2f, 15, 5, 0.7};
    case DDPMIR_ARCH_JPEG: return {4, 8, 4, 0.2f, 2.0f, 0.1f, 0.2f, 20, 5, 0.7};
    case DDPMIR_ARCH_AVIF: return {8, 8, 0, 0.f, 0.f, 0.15f, 0.15f, 30, 3, 0.8};
    default: throw std::runtime_error("unknown arch");
  }
}

// DDPMIR_DEBUG_SYNC=1 (diagnostics, read once): synchronise after every stage of a forward and name it on stderr, so that a
// GPU memory-access fault -- reported asynchronously, without a kernel name -- is pinned to the stage that was running.
bool debug_sync_enabled() {
  static const bool on = [] { const char* e = getenv("DDPMIR_DEBUG_SYNC"); return e && *e && *e != '0'; }();
  return on;
}
void debug_stage(hipStream_t s, const char* block, const char* what, int B, int H, int W, int C) {
  if (!debug_sync_enabled()) return;
  fprintf(stderr, "[ddpmir] %s %s B=%d %dx%d C=%d ...", block, what, B, H, W, C);
  fflush(stderr);
  const hipError_t e = hipStreamSynchronize(s);
  fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
  fflush(stderr);
}

}  // namespace

Context::Context(const ddpmir_config& c) : cfg(c), k(consts_for(c.arch)) {
  if (cfg.dtype != DDPMIR_F32 && cfg.dtype != DDPMIR_BF16 && cfg.dtype != DDPMIR_FP8)
    throw std::runtime_error("dtype must be DDPMIR_F32, DDPMIR_BF16 or DDPMIR_FP8");
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  DDPMIR_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  DDPMIR_HIP_CHECK(hipEventCreate(&ev0_));
  DDPMIR_HIP_CHECK(hipEventCreate(&ev1_));
  DDPMIR_HIP_CHECK(hipEventCreateWithFlags(&lane_ev_[0], hipEventDisableTiming));
  DDPMIR_HIP_CHECK(hipEventCreateWithFlags(&lane_ev_[1], hipEventDisableTiming));
  DDPMIR_HIP_CHECK(hipMalloc(&attn_general_taken_, 16));
  DDPMIR_HIP_CHECK(hipMemset(attn_general_taken_, 0, 16));
  for (auto& d : kBlocks) wt_.nt_total += d.cout;   // configuration constant: a plan made before the weights exist sizes tproj from it
}

Context::~Context() {
  (void)hipSetDevice(cfg.device);
  free_plan();
  if (wblob_ && wblob_owned_) (void)hipFree(wblob_);
  lane_.reset();
  for (auto e : lane_done_) if (e) (void)hipEventDestroy(e);
  if (attn_general_taken_) (void)hipFree(attn_general_taken_);
  if (pin_in_) (void)hipHostFree(pin_in_);
  if (pin_out_) (void)hipHostFree(pin_out_);
  for (auto& sp : prof_spans_) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  for (auto e : prof_pool_) (void)hipEventDestroy(e);
  for (auto e : lane_ev_) if (e) (void)hipEventDestroy(e);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// ------------------------------------------------------------------------------------------------ weights
void Context::load_tensor(const std::string& name, const float* data, const int64_t* shape, int ndim) {
  if (finalized_) throw std::runtime_error("load_tensor after finalize_weights");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  host_[name] = std::move(t);
}

void Context::put(Arena& a, bool upload, float*& dst, const std::string& name, size_t expect) {
  dst = a.take<float>(expect);
  if (name.find("dct_matrix") == std::string::npos) layout_params_ += (int64_t)expect;
  if (!upload) return;
  auto it = host_.find(name);
  if (it == host_.end()) throw std::runtime_error("missing tensor in state dict: " + name);
  if ((size_t)it->second.numel() != expect)
    throw std::runtime_error("shape mismatch for " + name + ": expected " + std::to_string(expect) + " elements, got " +
                             std::to_string(it->second.numel()));
  DDPMIR_HIP_CHECK(hipMemcpy(dst, it->second.data.data(), expect * sizeof(float), hipMemcpyHostToDevice));
}

// torch OIHW (or [out][in] for nn.Linear) -> [tap][Cout][CinP], CinP = Cin rounded up to 4 (zeros)
void Context::put_conv(Arena& a, bool upload, ConvW& cw, const std::string& name, int cout, int cin, int ksz) {
  cw.cout = cout; cw.cin = cin; cw.cinp = round_up(cin, 4); cw.taps = ksz * ksz;
  // the DMA GEMM wants whole 16-byte channel chunks and float4 output columns: its bf16 copy pads Cin to 8, Cout to 4
  cw.cin16 = use_gemm16() ? round_up(cin, 8) : cw.cinp;
  cw.cout16 = use_gemm16() ? round_up(cout, 4) : cout;
  const size_t n = (size_t)cw.taps * cout * cw.cinp, n16 = (size_t)cw.taps * cw.cout16 * cw.cin16;
  cw.w = a.take<float>(n);
  cw.b = a.take<float>(cw.cout16);
  cw.w16 = use_gemm16() ? a.take<unsigned short>(n16) : nullptr;
  const bool f8 = cfg.dtype == DDPMIR_FP8 && ksz == 3 && cin >= fp8_min_ch_ && cout >= fp8_min_ch_ && (cin % 16) == 0 && cout > 32;
  cw.is_f8 = f8;
  cw.w8 = f8 ? a.take<unsigned char>(n16) : nullptr;
  cw.w8_inv_dev = f8 ? a.take<float>(1) : nullptr;
  layout_params_ += (int64_t)cout * cin * cw.taps + cout;
  if (!upload) return;
  auto iw = host_.find(name + ".weight");
  if (iw == host_.end()) iw = host_.find(name + "_weight");   // attn.in_proj_weight
  auto ib = host_.find(name + ".bias");
  if (ib == host_.end()) ib = host_.find(name + "_bias");
  if (iw == host_.end() || ib == host_.end()) throw std::runtime_error("missing tensor in state dict: " + name);
  if (iw->second.numel() != (int64_t)cout * cin * ksz * ksz || ib->second.numel() != cout)
    throw std::runtime_error("shape mismatch for " + name);
  std::vector<float> packed(n, 0.f);
  const float* src = iw->second.data.data();
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < cw.taps; ++t) packed[((size_t)t * cout + o) * cw.cinp + c] = src[((size_t)o * cin + c) * cw.taps + t];
  DDPMIR_HIP_CHECK(hipMemcpy(cw.w, packed.data(), n * sizeof(float), hipMemcpyHostToDevice));
  if (cw.w16) {   // round-to-nearest-even bf16 copy for the matrix cores
    std::vector<unsigned short> p16(n16, 0);
    for (int t = 0; t < cw.taps; ++t)
      for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c) {
          unsigned int u;
          std::memcpy(&u, &packed[((size_t)t * cout + o) * cw.cinp + c], 4);
          p16[((size_t)t * cw.cout16 + o) * cw.cin16 + c] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
        }
    DDPMIR_HIP_CHECK(hipMemcpy(cw.w16, p16.data(), n16 * 2, hipMemcpyHostToDevice));
  }
  if (cw.w8) {   // e4m3 copy with a per-tensor power-of-two scale that uses the format's range (|w| <= 448 after scaling)
    float amax = 0.f;
    for (float v : packed) amax = std::max(amax, std::fabs(v));
    const int kexp = amax > 0.f ? (int)std::floor(std::log2(448.0 / (double)amax)) : 0;
    const float sw = std::ldexp(1.0f, kexp), inv = std::ldexp(1.0f, -kexp);
    cw.w8_kexp = kexp;
    std::vector<unsigned char> p8(n16, 0);
    for (int t = 0; t < cw.taps; ++t)
      for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
          p8[((size_t)t * cw.cout16 + o) * cw.cin16 + c] = f32_to_e4m3(packed[((size_t)t * cout + o) * cw.cinp + c] * sw);
    DDPMIR_HIP_CHECK(hipMemcpy(cw.w8, p8.data(), n16, hipMemcpyHostToDevice));
    DDPMIR_HIP_CHECK(hipMemcpy(cw.w8_inv_dev, &inv, 4, hipMemcpyHostToDevice));
  }
  std::vector<float> bias(cw.cout16, 0.f);
  std::memcpy(bias.data(), ib->second.data.data(), cout * sizeof(float));
  DDPMIR_HIP_CHECK(hipMemcpy(cw.b, bias.data(), cw.cout16 * sizeof(float), hipMemcpyHostToDevice));
}

// Frequency gates for the DMA GEMM: the low and the high gate (each C -> C/2 -> C, webp_inference.py:214-227) act on
// complementary pixel sets, so both first layers are stacked into one [C][C] GEMM on the unmasked DCT tensor and both
// second layers into one block-diagonal [2C][C] GEMM whose epilogue keeps, per pixel, the half that owns it.
void Context::put_gate_pair(Arena& a, bool upload, BlockW& b, const std::string& fg) {
  const int C = b.cout, Hd = C / 2;
  auto init = [&](ConvW& cw, int cout, int cin) {
    cw.cout = cw.cout16 = cout; cw.cin = cw.cinp = cw.cin16 = cin; cw.taps = 1;
    cw.w = nullptr;
    cw.b = a.take<float>(cout);
    cw.w16 = a.take<unsigned short>((size_t)cout * cin);
  };
  init(b.fg1, C, C);
  init(b.fg2, 2 * C, C);
  if (!upload) return;
  auto get = [&](const std::string& n, int64_t numel) -> const float* {
    auto it = host_.find(n);
    if (it == host_.end()) throw std::runtime_error("missing tensor in state dict: " + n);
    if (it->second.numel() != numel) throw std::runtime_error("shape mismatch for " + n);
    return it->second.data.data();
  };
  const float* wl0 = get(fg + ".low_freq_attn.0.weight", (int64_t)Hd * C);
  const float* wh0 = get(fg + ".high_freq_attn.0.weight", (int64_t)Hd * C);
  const float* wl2 = get(fg + ".low_freq_attn.2.weight", (int64_t)C * Hd);
  const float* wh2 = get(fg + ".high_freq_attn.2.weight", (int64_t)C * Hd);
  const float* bl0 = get(fg + ".low_freq_attn.0.bias", Hd);
  const float* bh0 = get(fg + ".high_freq_attn.0.bias", Hd);
  const float* bl2 = get(fg + ".low_freq_attn.2.bias", C);
  const float* bh2 = get(fg + ".high_freq_attn.2.bias", C);
  auto bf = [](float f) {
    unsigned int u;
    std::memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  };
  std::vector<unsigned short> w1((size_t)C * C), w2((size_t)2 * C * C, 0);
  std::vector<float> b1(C), b2v(2 * C);
  for (int o = 0; o < Hd; ++o)
    for (int c = 0; c < C; ++c) { w1[(size_t)o * C + c] = bf(wl0[(size_t)o * C + c]); w1[(size_t)(Hd + o) * C + c] = bf(wh0[(size_t)o * C + c]); }
  for (int o = 0; o < Hd; ++o) { b1[o] = bl0[o]; b1[Hd + o] = bh0[o]; }
  for (int o = 0; o < C; ++o)
    for (int c = 0; c < Hd; ++c) {
      w2[(size_t)o * C + c] = bf(wl2[(size_t)o * Hd + c]);                 // low gate reads hidden [0, C/2)
      w2[(size_t)(C + o) * C + Hd + c] = bf(wh2[(size_t)o * Hd + c]);      // high gate reads hidden [C/2, C)
    }
  for (int o = 0; o < C; ++o) { b2v[o] = bl2[o]; b2v[C + o] = bh2[o]; }
  DDPMIR_HIP_CHECK(hipMemcpy(b.fg1.w16, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
  DDPMIR_HIP_CHECK(hipMemcpy(b.fg2.w16, w2.data(), w2.size() * 2, hipMemcpyHostToDevice));
  DDPMIR_HIP_CHECK(hipMemcpy(b.fg1.b, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
  DDPMIR_HIP_CHECK(hipMemcpy(b.fg2.b, b2v.data(), b2v.size() * 4, hipMemcpyHostToDevice));
}

void Context::layout_weights(Arena& a, bool up, Weights& w) {
  // every pointer goes into the table `w` the caller handed in -- never through the members (see model.h)
  layout_params_ = 0;
  auto& blocks = w.blocks;
  blocks.assign(13, BlockW());
  float *w0, *b0, *w2, *b2;
  put(a, up, w0, "time_embed.proj.0.weight", 1024 * 256);
  put(a, up, b0, "time_embed.proj.0.bias", 1024);
  put(a, up, w2, "time_embed.proj.2.weight", 256 * 1024);
  put(a, up, b2, "time_embed.proj.2.bias", 256);
  w.nt_total = 0;
  for (auto& d : kBlocks) w.nt_total += d.cout;
  float* wt = a.take<float>((size_t)w.nt_total * 256);
  float* bt = a.take<float>(w.nt_total);
  w.tw = TimeEmbedW{w0, b0, w2, b2, wt, bt, w.nt_total};
  int off = 0;
  const bool avif = cfg.arch == DDPMIR_ARCH_AVIF;
  for (int i = 0; i < 13; ++i) {
    const BlockDef& d = kBlocks[i];
    BlockW& b = blocks[i];
    const std::string n = d.name;
    b.name = n; b.cin = d.cin; b.cinp = round_up(d.cin, 4); b.cout = d.cout; b.level = d.level;
    b.has_shortcut = d.cin != d.cout;
    b.tproj_off = off;
    put(a, up, b.n1g, n + ".norm1.weight", d.cin);
    put(a, up, b.n1b, n + ".norm1.bias", d.cin);
    put_conv(a, up, b.conv1, n + ".conv1", d.cout, d.cin, 3);
    if (up) {
      auto iw = host_.find(n + ".time_proj.weight"), ib = host_.find(n + ".time_proj.bias");
      if (iw == host_.end() || ib == host_.end()) throw std::runtime_error("missing tensor in state dict: " + n + ".time_proj");
      if (iw->second.numel() != (int64_t)d.cout * 256 || ib->second.numel() != d.cout)
        throw std::runtime_error("shape mismatch for " + n + ".time_proj");
      DDPMIR_HIP_CHECK(hipMemcpy(wt + (size_t)off * 256, iw->second.data.data(), (size_t)d.cout * 256 * 4, hipMemcpyHostToDevice));
      DDPMIR_HIP_CHECK(hipMemcpy(bt + off, ib->second.data.data(), (size_t)d.cout * 4, hipMemcpyHostToDevice));
    }
    layout_params_ += (int64_t)d.cout * 257;
    off += d.cout;
    put(a, up, b.n2g, n + ".norm2.weight", d.cout);
    put(a, up, b.n2b, n + ".norm2.bias", d.cout);
    put_conv(a, up, b.conv2, n + ".conv2", d.cout, d.cout, 3);
    put_conv(a, up, b.in_proj, n + ".attn.in_proj", 3 * d.cout, d.cout, 1);
    put_conv(a, up, b.out_proj, n + ".attn.out_proj", d.cout, d.cout, 1);
    const std::string fg = n + ".freq_guide";
    if (avif) {
      put(a, up, b.at.tw, fg + ".adaptive_transform.transform_weights", (size_t)d.cout * 64);
      // inverse_weights is a parameter the forward never reads (avif_inference.py:130): validated and counted only
      if (up && host_.find(fg + ".adaptive_transform.inverse_weights") == host_.end())
        throw std::runtime_error("missing tensor in state dict: " + fg + ".adaptive_transform.inverse_weights");
      layout_params_ += (int64_t)d.cout * 64;
      put_conv(a, up, b.at.q0, fg + ".adaptive_transform.quantization.0", d.cout, d.cout, 1);
      put_conv(a, up, b.at.q2, fg + ".adaptive_transform.quantization.2", d.cout, d.cout, 1);
      for (int s = 0; s < 4; ++s) {
        put_conv(a, up, b.ms1[s], fg + ".multi_scale_attn." + std::to_string(s) + ".1", d.cout / 4, d.cout, 1);
        put_conv(a, up, b.ms3[s], fg + ".multi_scale_attn." + std::to_string(s) + ".3", d.cout, d.cout / 4, 1);
      }
      put_conv(a, up, b.cc0, fg + ".color_consistency.0", d.cout, d.cout, 1);
      put_conv(a, up, b.cc2, fg + ".color_consistency.2", d.cout, d.cout, 1);
      put_conv(a, up, b.ep0, fg + ".edge_preserve.0", d.cout / 2, d.cout, 3);
      put_conv(a, up, b.ep2, fg + ".edge_preserve.2", d.cout, d.cout / 2, 3);
    } else {
      put(a, up, b.dct, fg + ".dct.dct_matrix", (size_t)k.dct_block * k.dct_block);
      put_conv(a, up, b.low0, fg + ".low_freq_attn.0", d.cout / 2, d.cout, 1);
      put_conv(a, up, b.low2, fg + ".low_freq_attn.2", d.cout, d.cout / 2, 1);
      put_conv(a, up, b.high0, fg + ".high_freq_attn.0", d.cout / 2, d.cout, 1);
      put_conv(a, up, b.high2, fg + ".high_freq_attn.2", d.cout, d.cout / 2, 1);
      if (use_gemm16()) put_gate_pair(a, up, b, fg);
    }
    put_conv(a, up, b.conv_out, fg + ".conv_out", d.cout, d.cout, 3);
    if (b.has_shortcut) put_conv(a, up, b.shortcut, n + ".shortcut", d.cout, d.cin, 1);
    if (b.has_shortcut && b.conv_out.is_f8) {   // bf16 shortcut weights on conv_out's fp8 scale (second K segment of the e4m3 conv_out)
      const size_t nsc = (size_t)b.shortcut.cout16 * b.shortcut.cin16;
      b.sc16s = a.take<unsigned short>(nsc);
      if (up) {
        const auto& w = host_.at(std::string(d.name) + ".shortcut.weight").data;   // [cout][cin] (1x1)
        const float sw = std::ldexp(1.0f, b.conv_out.w8_kexp);
        std::vector<unsigned short> ws(nsc, 0);
        for (int o = 0; o < d.cout; ++o)
          for (int c = 0; c < d.cin; ++c) {
            const float f = w[(size_t)o * d.cin + c] * sw;
            unsigned int u;
            std::memcpy(&u, &f, 4);
            ws[(size_t)o * b.shortcut.cin16 + c] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
          }
        DDPMIR_HIP_CHECK(hipMemcpy(b.sc16s, ws.data(), nsc * 2, hipMemcpyHostToDevice));
      }
    }
    if (b.has_shortcut && use_gemm16()) {
      b.co_bias = a.take<float>(d.cout);
      if (up) {
        const auto& b1 = host_.at(fg + ".conv_out.bias").data;
        const auto& b2 = host_.at(n + ".shortcut.bias").data;
        std::vector<float> sum(d.cout);
        for (int i = 0; i < d.cout; ++i) sum[i] = b1[i] + b2[i];
        DDPMIR_HIP_CHECK(hipMemcpy(b.co_bias, sum.data(), sum.size() * 4, hipMemcpyHostToDevice));
      }
    }
  }
  if (avif) {
    put(a, up, w.head_at.tw, "avif_layer.transform_weights", 64 * 64);
    if (up && host_.find("avif_layer.inverse_weights") == host_.end())
      throw std::runtime_error("missing tensor in state dict: avif_layer.inverse_weights");
    layout_params_ += 64 * 64;
    put_conv(a, up, w.head_at.q0, "avif_layer.quantization.0", 64, 64, 1);
    put_conv(a, up, w.head_at.q2, "avif_layer.quantization.2", 64, 64, 1);
  } else {
    put(a, up, w.head_dct, "dct_layer.dct_matrix", (size_t)k.dct_block * k.dct_block);
  }
  put(a, up, w.head_g, "out_conv.0.weight", 64);
  put(a, up, w.head_b, "out_conv.0.bias", 64);
  put_conv(a, up, w.head_conv, "out_conv.2", 3, 64, 3);
  w.n_params = layout_params_;
}

size_t Context::weight_blob_bytes() {
  // The size comes from a DRY RUN of layout_weights over an arena with no base address.  Until round 3 that dry run wrote the
  // context's own pointer table, i.e. re-pointed every weight at NULL + offset: harmless before finalize_weights, a memory
  // access fault after it ("... on address 0x31000": bench.py asked for the blob size between two sampling calls).  Now the
  // dry run fills a scratch table that is thrown away, so a size query can never touch the live one, before or after
  // finalize; the size is still cached per configuration (set_option("fp8_min_channels") resets it).
  if (blob_bytes_ == 0) {
    Weights scratch;
    Arena a;
    layout_weights(a, false, scratch);
    blob_bytes_ = (a.off + 255) & ~size_t(255);
    if (!finalized_) wt_.n_params = scratch.n_params;   // param_count() before the weights are loaded
  }
  return blob_bytes_;
}

void Context::finalize_weights(void* dev_blob, int fill) {
  if (finalized_) throw std::logic_error("finalize_weights called twice on one context");
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  free_plan();   // a workspace planned before the weights existed (jpeg_roundtrip / phase_consistency) is re-planned on first use
  const size_t bytes = weight_blob_bytes();
  if (dev_blob) {
    wblob_ = static_cast<char*>(dev_blob);
    wblob_owned_ = false;
  } else {
    DDPMIR_HIP_CHECK(hipMalloc(&wblob_, bytes));
    wblob_owned_ = true;
    fill = 1;
  }
  Arena a;
  a.base = wblob_;
  if (fill) DDPMIR_HIP_CHECK(hipMemset(wblob_, 0, bytes));
  layout_weights(a, fill != 0, wt_);
  DDPMIR_HIP_CHECK(hipDeviceSynchronize());
  if (cfg.dtype == DDPMIR_FP8)   // the fp8 weight scales travel inside the blob (adopted blobs carry them too)
    for (auto& b : blocks_)
      for (ConvW* cw : {&b.conv1, &b.conv2, &b.conv_out, &b.ep0, &b.ep2})
        if (cw->w8) DDPMIR_HIP_CHECK(hipMemcpy(&cw->w8_inv, cw->w8_inv_dev, 4, hipMemcpyDeviceToHost));
  host_.clear();
  finalized_ = true;
}

// ------------------------------------------------------------------------------------------------ workspace
void Context::layout_plan(Arena& a, Plan& p, int B, int H, int W) const {
  const bool avif = cfg.arch == DDPMIR_ARCH_AVIF;
  auto hw = [&](int lvl) { return (size_t)(H >> lvl) * (W >> lvl); };
  const size_t M0 = (size_t)B * H * W;
  p.x4 = a.take<float>(M0 * 4);
  // the activation stream between blocks: bf16 tensors where stream16(), fp32 otherwise
  const bool h = stream16();
  auto act = [&](size_t pixels, int c) {
    ActBuf t; t.ld = c; t.h = h;
    if (h) t.p = a.take<unsigned short>(pixels * c); else t.p = a.take<float>(pixels * c);
    return t;
  };
  const int cat_c[5] = {128, 256, 512, 1024, 1024};            // concat width at level 0..4
  for (int l = 0; l < 5; ++l) p.cat[l] = act((size_t)B * hw(l), cat_c[l]);
  const int pin_c[6] = {0, 64, 128, 256, 512, 512};            // pooled input channels at level 1..5
  for (int l = 1; l <= 5; ++l) p.pin[l] = act((size_t)B * hw(l), pin_c[l]);
  p.bb0 = act((size_t)B * hw(5), 1024);
  p.bb1 = act((size_t)B * hw(5), 1024);
  p.bb2 = act((size_t)B * hw(5), 512);
  const int u_c[5] = {64, 64, 128, 256, 512};                  // up-block output channels at level 0..4
  for (int l = 0; l < 5; ++l) p.u[l] = act((size_t)B * hw(l), u_c[l]);
  // (up5's output joins the bf16 stream too: the head's mixed transform goes through an LDS tile with 16-byte I/O, head_mix16_k;
  //  a thread-per-channel pass with 2-byte global accesses ran 4x slower than its fp32 twin)
  // per-block scratch: max over blocks of pixels*cout is at level 0 (64 ch); deeper levels never exceed it
  size_t mc = 0;
  for (auto& d : kBlocks) mc = std::max(mc, (size_t)B * hw(d.level) * d.cout);
  p.hA = a.take<float>(mc); p.hB = a.take<float>(mc); p.att = a.take<float>(mc);
  p.xd = a.take<float>(mc); p.S = a.take<float>(mc); p.hid = a.take<float>(mc);
  p.qkv = a.take<float>(mc * 3);
  if (avif) {
    p.xt = a.take<float>(mc); p.avg = a.take<float>(mc); p.col = a.take<float>(mc); p.edge = a.take<float>(mc);
    p.pool = a.take<float>((size_t)B * 85 * 1024); p.pg1 = a.take<float>((size_t)B * 85 * 256);   // all four scales side by side
    p.pool8 = a.take<float>((size_t)B * 64 * 1024);
    p.pg2 = a.take<float>((size_t)B * 85 * 1024);   // the four pooled gate maps (1 + 4 + 16 + 64 positions) back to back
  }
  if (use_gemm16()) {   // bf16 attention operand images (bf16 and fp8 modes)
    size_t ws = 0;
    for (auto& d : kBlocks)
      if (d.level >= cfg.attn_min_level && attention_bf16_supported((int)hw(d.level), d.cout, k.heads))
        ws = std::max(ws, attention_bf16_workspace(B, (int)hw(d.level), d.cout, k.heads));
    p.attn_ws = a.take<unsigned char>(ws + 256);
  }
  if (use_gemm16()) {
    size_t ma = 0;
    for (auto& d : kBlocks) ma = std::max(ma, (size_t)B * hw(d.level) * (size_t)std::max(round_up(d.cin, 8), d.cout));
    p.a16 = a.take<unsigned short>(ma); p.x16 = a.take<unsigned short>(ma);
    p.hB16 = a.take<unsigned short>(mc); p.att16 = a.take<unsigned short>(mc); p.xdl16 = a.take<unsigned short>(mc);
    p.xdh16 = a.take<unsigned short>(mc); p.hid16 = a.take<unsigned short>(mc); p.S16 = a.take<unsigned short>(mc);
    if (cfg.dtype == DDPMIR_FP8) {
      size_t m8 = 0;   // largest e4m3 operand: pixels * channels over the blocks that have fp8 convs (C >= 256: levels >= 2)
      for (auto& d : kBlocks)
        if (d.cout >= fp8_min_ch_) m8 = std::max(m8, (size_t)B * hw(d.level) * (size_t)std::max(d.cin, d.cout));
      p.a8 = a.take<unsigned char>(m8 + 16); p.S8 = a.take<unsigned char>(m8 + 16);
    }
    p.zero16 = a.take<unsigned char>(GEMM16_ZERO_BYTES);
    p.splitk_ws = a.take<float>(GEMM16_SPLITK_WS_BYTES / sizeof(float));
  }
  int max_splits = gn_splits(H * W, 64);
  p.gn_partial = a.take<double>((size_t)B * 8 * max_splits * 2 + 64);
  p.gn_sums = a.take<double>(2 * gn_sums_doubles(B, H * W, 8) + 64);   // (x2: the head's per-4x4-block partials, H*W/16 slots)
  for (int l = 0; l < 5; ++l) p.skip_sums[l] = use_gemm16() ? a.take<double>(gn_sums_doubles(B, (int)hw(l), 4) + 64) : nullptr;
  p.sc1 = a.take<float>((size_t)B * 1024); p.sh1 = a.take<float>((size_t)B * 1024);
  p.sc2 = a.take<float>((size_t)B * 1024); p.sh2 = a.take<float>((size_t)B * 1024);
  p.hidden = a.take<float>((size_t)B * 1024); p.temb = a.take<float>((size_t)B * 256);
  p.tproj = a.take<float>((size_t)B * nt_total_);
  p.boost = a.take<float>(B); p.boost2 = a.take<float>(B); p.constmix = a.take<float>(B);
  // sampler
  const size_t n_img = (size_t)B * 3 * H * W;
  p.s_xt = a.take<float>(n_img); p.s_theta = a.take<float>(n_img); p.s_y = a.take<float>(n_img);
  p.s_noise = a.take<float>(n_img + 4); p.s_t = a.take<float>(B);
  p.sp = a.take<StepParams>(2);
  p.s_u8 = a.take<uint8_t>(n_img); p.s_dec = a.take<uint8_t>(n_img);
  p.s_jpeg = a.take<uint8_t>(jpeg_sim_workspace(B, H, W));
  for (int i = 0; i < 10; ++i) p.dft[i] = a.take<float>(n_img);
  p.tw_ch = a.take<float>(H); p.tw_sh = a.take<float>(H); p.tw_cw = a.take<float>(W); p.tw_sw = a.take<float>(W);
  p.tw64_ch = a.take<double>(H); p.tw64_sh = a.take<double>(H); p.tw64_cw = a.take<double>(W); p.tw64_sw = a.take<double>(W);
  p.dft64_re = a.take<double>(n_img); p.dft64_im = a.take<double>(n_img);
  p.g_x = a.take<float>(n_img); p.g_out = a.take<float>(n_img); p.g_t = a.take<float>(B); p.g_cl = a.take<float>(B);
}

void Context::free_plan() {
  drop_loop_graphs();
  if (plan_.gexec) (void)hipGraphExecDestroy(plan_.gexec);
  if (plan_.graph) (void)hipGraphDestroy(plan_.graph);
  if (plan_.base) (void)hipFree(plan_.base);
  plan_ = Plan();
}

void Context::ensure_plan(int B, int H, int W) {
  if (plan_.base && plan_.B == B && plan_.H == H && plan_.W == W) return;
  if (B <= 0 || H <= 0 || W <= 0 || (H % 32) != 0 || (W % 32) != 0)
    throw std::invalid_argument("image size must be a positive multiple of 32 (five 2x2 pools)");
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  free_plan();
  size_t bytes;
  {   // sizing pass: a scratch Plan over a base-less arena -- plan_ stays empty until it can be laid out over real memory
    Plan scratch;
    Arena dry;
    layout_plan(dry, scratch, B, H, W);
    bytes = (dry.off + 255) & ~size_t(255);
  }
  char* base = nullptr;
  DDPMIR_HIP_CHECK(hipMalloc(&base, bytes));
  DDPMIR_HIP_CHECK(hipMemset(base, 0, bytes));
  Arena a; a.base = base;
  layout_plan(a, plan_, B, H, W);
  plan_.base = base; plan_.bytes = bytes; plan_.B = B; plan_.H = H; plan_.W = W;
  fb_ = B;
  // DFT twiddles (double precision on the host, rounded once)
  auto tw = [&](int n, float* dc, float* ds) {
    std::vector<float> c(n), s(n);
    for (int i = 0; i < n; ++i) {
      double ang = 2.0 * M_PI * (double)i / (double)n;
      c[i] = (float)std::cos(ang); s[i] = (float)std::sin(ang);
    }
    DDPMIR_HIP_CHECK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice));
    DDPMIR_HIP_CHECK(hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice));
  };
  tw(H, plan_.tw_ch, plan_.tw_sh);
  tw(W, plan_.tw_cw, plan_.tw_sw);
  auto tw64 = [&](int n, double* dc, double* ds) {
    std::vector<double> c(n), s(n);
    for (int i = 0; i < n; ++i) {
      // exact at the multiples of a quarter turn, so purely real / imaginary bins stay exactly so
      const int q = (4 * i) % n == 0 ? (4 * i) / n : -1;
      const double ang = 2.0 * M_PI * (double)i / (double)n;
      c[i] = q == 0 ? 1.0 : q == 1 ? 0.0 : q == 2 ? -1.0 : q == 3 ? 0.0 : std::cos(ang);
      s[i] = q == 0 ? 0.0 : q == 1 ? 1.0 : q == 2 ? 0.0 : q == 3 ? -1.0 : std::sin(ang);
    }
    DDPMIR_HIP_CHECK(hipMemcpy(dc, c.data(), n * 8, hipMemcpyHostToDevice));
    DDPMIR_HIP_CHECK(hipMemcpy(ds, s.data(), n * 8, hipMemcpyHostToDevice));
  };
  tw64(H, plan_.tw64_ch, plan_.tw64_sh);
  tw64(W, plan_.tw64_cw, plan_.tw64_sw);
  std::vector<float> mix(B, k.head_mix);
  DDPMIR_HIP_CHECK(hipMemcpy(plan_.constmix, mix.data(), B * 4, hipMemcpyHostToDevice));
  size_t need = (size_t)B * 3 * H * W;
  if (need > pin_bytes_) {
    if (pin_in_) (void)hipHostFree(pin_in_);
    if (pin_out_) (void)hipHostFree(pin_out_);
    DDPMIR_HIP_CHECK(hipHostMalloc(&pin_in_, need));
    DDPMIR_HIP_CHECK(hipHostMalloc(&pin_out_, need));
    pin_bytes_ = need;
  }
}

// ------------------------------------------------------------------------------------------------ profiling
enum { PROF_ATTN_L0 = 0, PROF_ATTN = 1, PROF_FWD = 2, PROF_GEMM = 3, PROF_GEMM8 = 4 };

hipEvent_t Context::prof_event() {
  if (!prof_pool_.empty()) { hipEvent_t e = prof_pool_.back(); prof_pool_.pop_back(); return e; }
  hipEvent_t e;
  DDPMIR_HIP_CHECK(hipEventCreate(&e));
  return e;
}
void Context::prof_begin(int cls, double flops, hipStream_t s) {
  prof_open_ = false;
  // the coarse level (1) brackets the forward and its two full-resolution attention launches only -- 6 event records per
  // forward instead of 54 -- so that the measured stream carries as few markers as the roofline figures need
  if (!prof_on_ || (cls == PROF_ATTN && prof_detail_ < 2) || ((cls == PROF_GEMM || cls == PROF_GEMM8) && prof_detail_ < 3)) return;
  ProfSpan sp{prof_event(), prof_event(), cls, flops};
  DDPMIR_HIP_CHECK(hipEventRecord(sp.a, s));
  prof_spans_.push_back(sp);
  prof_open_ = true;
}
void Context::prof_end(hipStream_t s) {
  if (!prof_on_ || !prof_open_) return;
  DDPMIR_HIP_CHECK(hipEventRecord(prof_spans_.back().b, s));
  prof_open_ = false;
}
void Context::prof_collect() {
  for (auto& sp : prof_spans_) {
    DDPMIR_HIP_CHECK(hipEventSynchronize(sp.b));
    float ms = 0.f;
    DDPMIR_HIP_CHECK(hipEventElapsedTime(&ms, sp.a, sp.b));
    prof_ms_[sp.cls] += ms; prof_n_[sp.cls] += sp.count; prof_flops_[sp.cls] += sp.flops;
    if (sp.cls == PROF_ATTN_L0) { prof_ms_[PROF_ATTN] += ms; prof_n_[PROF_ATTN] += 1; prof_flops_[PROF_ATTN] += sp.flops; }
    prof_pool_.push_back(sp.a); prof_pool_.push_back(sp.b);
  }
  prof_spans_.clear();
}
void Context::profile_enable(int enable) {
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  if (enable) {
    // (with use_graph = 1 the loop and predict() launch eagerly while profiling is on: events cannot bracket graph nodes)
    prof_collect();
    for (int i = 0; i < kProfClasses; ++i) { prof_ms_[i] = 0; prof_n_[i] = 0; prof_flops_[i] = 0; }
    codec_host_ms_ = 0.0; codec_calls_ = 0;
    DDPMIR_HIP_CHECK(hipStreamSynchronize(stream_));
    DDPMIR_HIP_CHECK(hipMemset(attn_general_taken_, 0, 16));
    attn_general_total_ = 0;
  }
  prof_on_ = enable != 0;
  prof_detail_ = enable;
  if (lane_) lane_->profile_enable(enable);
}
void Context::profile_read(const std::string& cls, double* ms, int64_t* launches, double* flops) {
  if (lane_) {   // two concurrent lanes: a class is the sum over both lanes' launches
    double m0 = 0, f0 = 0, m1 = 0, f1 = 0; int64_t n0 = 0, n1 = 0;
    std::unique_ptr<Context> keep = std::move(lane_);
    profile_read(cls, &m0, &n0, &f0);
    lane_ = std::move(keep);
    lane_->profile_read(cls, &m1, &n1, &f1);
    if (ms) *ms = m0 + m1;
    if (launches) *launches = n0 + n1;
    if (flops) *flops = f0 + f1;
    return;
  }
  prof_collect();
  if (cls == "codec_host") {   // host milliseconds inside the codec callback of the sampling loop, and its call count
    if (ms) *ms = codec_host_ms_;
    if (launches) *launches = codec_calls_;
    if (flops) *flops = 0.0;
    return;
  }
  if (cls == "attention_general") {   // launches = general-kernel workgroups that had work, flops = those launched beside a no-max kernel
    unsigned taken = 0;
    DDPMIR_HIP_CHECK(hipStreamSynchronize(stream_));
    DDPMIR_HIP_CHECK(hipMemcpy(&taken, attn_general_taken_, 4, hipMemcpyDeviceToHost));
    if (ms) *ms = 0.0;
    if (launches) *launches = (int64_t)taken;
    if (flops) *flops = (double)attn_general_total_;
    return;
  }
  int c = cls == "attention_l0" ? PROF_ATTN_L0 : cls == "attention" ? PROF_ATTN : cls == "forward" ? PROF_FWD :
          cls == "gemm" ? PROF_GEMM : cls == "gemm_fp8" ? PROF_GEMM8 : -1;
  if (c < 0) throw std::invalid_argument("unknown profile class " + cls);
  if (ms) *ms = prof_ms_[c];
  if (launches) *launches = prof_n_[c];
  if (flops) *flops = prof_flops_[c];
}

AttnOpts Context::attn_opts() {
  AttnOpts o;
  o.nm_bound = attn_nm_bound_;
  if (prof_on_) { o.general_taken = attn_general_taken_; o.general_total = &attn_general_total_; }
  return o;
}

void Context::set_option(const std::string& name, double value) {
  lane_.reset();   // the second lane's context copies the options when it is created (loop_two_lanes)
  if (name == "concurrent_lanes") {   // -1 auto, 0 never, 1 whenever the loop allows it (see use_two_lanes)
    if (value != -1 && value != 0 && value != 1 && value != 2) throw std::invalid_argument("concurrent_lanes must be -1, 0, 1 or 2");
    lanes_opt_ = (int)value;
    return;
  }
  if (name == "attention_no_max_bound") {   // log2 units; < 0: every query block takes the general (running-max) kernel
    if (!(value <= 80.0)) throw std::invalid_argument("attention_no_max_bound must be <= 80 (fp32 range of 2^s sums)");
    attn_nm_bound_ = (float)value;
    drop_loop_graphs();                     // captured timesteps bake the bound in
    if (plan_.gexec) { (void)hipGraphExecDestroy(plan_.gexec); plan_.gexec = nullptr; }
    if (plan_.graph) { (void)hipGraphDestroy(plan_.graph); plan_.graph = nullptr; }
    return;
  }
  if (name == "fp8_min_channels") {   // DDPMIR_FP8: which 3x3 convs take e4m3 operands (before the weights are loaded: it sizes the blob)
    if (finalized_) throw std::logic_error("fp8_min_channels must be set before finalize_weights");
    if (value != 64 && value != 128 && value != 256) throw std::invalid_argument("fp8_min_channels must be 64, 128 or 256");
    fp8_min_ch_ = (int)value;
    blob_bytes_ = 0;
    free_plan();   // the plan sizes a8 / S8 from this option, and ensure_plan is reachable before finalize (jpeg_roundtrip, phase_consistency)
    return;
  }
  if (name == "bf16_activation_stream") {   // -1: fp8 mode only (default); 0: never; 1: bf16 mode too.  Re-plans (buffers change type)
    if (value != -1 && value != 0 && value != 1) throw std::invalid_argument("bf16_activation_stream must be -1, 0 or 1");
    stream16_opt_ = (int)value;
    free_plan();
    return;
  }
  if (name == "epilogue_direct") {   // 1 (default): register-direct epilogue of the 64-column GEMM kernels; 0: LDS-transposed everywhere (A/B, parity)
    if (value != 0 && value != 1) throw std::invalid_argument("epilogue_direct must be 0 or 1");
    epi_direct_opt_ = (int)value;
    drop_loop_graphs();
    if (plan_.gexec) { (void)hipGraphExecDestroy(plan_.gexec); plan_.gexec = nullptr; }
    if (plan_.graph) { (void)hipGraphDestroy(plan_.graph); plan_.graph = nullptr; }
    return;
  }
  if (name == "head_fused") {   // 1 (default): the output head in one pass (head_fused.hip); 0: GroupNorm-apply pass + strip conv + tanh pass
    if (value != 0 && value != 1) throw std::invalid_argument("head_fused must be 0 or 1");
    head_fused_opt_ = (int)value;
    drop_loop_graphs();
    if (plan_.gexec) { (void)hipGraphExecDestroy(plan_.gexec); plan_.gexec = nullptr; }
    if (plan_.graph) { (void)hipGraphDestroy(plan_.graph); plan_.graph = nullptr; }
    return;
  }
  if (name == "conv_tile") {   // GroupNorm-apply fused into the full-resolution 64 -> 64 3x3 convs (conv_tile.hip): 1 (default) from 1.5 M pixels, 2 always, 0 never
    if (value != 0 && value != 1 && value != 2) throw std::invalid_argument("conv_tile must be 0, 1 or 2");
    conv_tile_opt_ = (int)value;
    drop_loop_graphs();
    if (plan_.gexec) { (void)hipGraphExecDestroy(plan_.gexec); plan_.gexec = nullptr; }
    if (plan_.graph) { (void)hipGraphDestroy(plan_.graph); plan_.graph = nullptr; }
    return;
  }
  if (name == "gemm256_tile") {   // the 256 x 256 eight-wave tile for the Cout >= 256 3x3 convs: -1 where it fills the chip (default), 0 never, 1 wherever supported
    if (value != -1 && value != 0 && value != 1) throw std::invalid_argument("gemm256_tile must be -1, 0 or 1");
    tile256_opt_ = (int)value;
    drop_loop_graphs();                     // captured timesteps bake the kernel choice in
    if (plan_.gexec) { (void)hipGraphExecDestroy(plan_.gexec); plan_.gexec = nullptr; }
    if (plan_.graph) { (void)hipGraphDestroy(plan_.graph); plan_.graph = nullptr; }
    return;
  }
  throw std::invalid_argument("unknown option " + name);
}

// ------------------------------------------------------------------------------------------------ forward
void Context::conv(const ConvW& cw, IGemm p, hipStream_t s) {
  if (cw.w16 && (cw.cin16 != cw.cinp || cw.cout16 != cw.cout)) throw std::logic_error("conv: bf16 copy is packed for the DMA GEMM");
  p.Wt = cw.w; p.Wt16 = cw.w16; p.bias = cw.b; p.Cout = cw.cout; p.Cin = cw.cinp; p.taps = cw.taps;
  launch_igemm(p, s);
}

// WebP/JPEG frequency block (webp_inference.py:231-270).  In: plan_.hB.  Out: plan_.S = x + ga*low + gb*high.
void Context::freq_dct(const BlockW& bw, int H, int W, hipStream_t s) {
  Plan& p = plan_;
  const int B = fb_, C = bw.cout;
  launch_block_transform(p.hB, C, p.xd, C, B, H, W, C, k.dct_block, bw.dct, 0, 0.f, 1.f, s);
  IGemm g;
  g.B = B; g.H = H; g.W = W; g.bs = k.dct_block; g.low = k.low_size;
  // low gate: hid = lrelu(W1 (xd*Mlow) + b1); S = hB + sigmoid(W2 hid + b2) * xd * Mlow
  IGemm a = g; a.A = p.xd; a.lda = C; a.pro_mask = MASK_LOW; a.epi_act = ACT_LRELU; a.out = p.hid; a.ldo = C / 2;
  conv(bw.low0, a, s);
  IGemm b2 = g; b2.A = p.hid; b2.lda = C / 2; b2.epi_act = ACT_SIGMOID; b2.mul = p.xd; b2.ldmul = C;
  b2.epi_mask = MASK_LOW; b2.res = p.hB; b2.ldres = C; b2.out = p.S; b2.ldo = C;
  conv(bw.low2, b2, s);
  // high gate (boosted by clamp(1-cl)) accumulates in place
  IGemm c = a; c.pro_mask = MASK_HIGH;
  conv(bw.high0, c, s);
  IGemm d = b2; d.epi_mask = MASK_HIGH; d.rowscale = p.boost; d.res = p.S; d.out = p.S;
  conv(bw.high2, d, s);
}

// AVIFAdaptiveTransform (avif_inference.py:140-182): out = rowscale * (T * sigmoid(q2(relu(q0(T))))) + res
void Context::avif_transform(const AvifTransformW& at, const float* x, int ldx, int C, int H, int W, float* T,
                             float* tmp, float* out, int ldo, const float* rowscale, const float* res, int ldres,
                             hipStream_t s) {
  const int B = fb_;
  IGemm g; g.B = B; g.H = H; g.W = W;
  if (use_gemm16()) {   // bf16 copy of the transformed tensor feeds the two 1x1 "quantization" GEMMs by LDS-DMA
    Plan& p = plan_;
    launch_block_transform(x, ldx, T, C, B, H, W, C, 8, at.tw, 1, 0.f, 1.f, s, p.xdl16, nullptr, 0);
    IGemm a = g; a.A16 = p.xdl16; a.lda16 = C; a.epi_act = ACT_RELU; a.out16 = p.hid16; a.ldo16 = C;
    conv16(at.q0, a, s);
    IGemm b = g; b.A16 = p.hid16; b.lda16 = C; b.epi_act = ACT_SIGMOID; b.mul = T; b.ldmul = C; b.rowscale = rowscale;
    b.res = res; b.ldres = ldres; b.out = out; b.ldo = ldo;
    conv16(at.q2, b, s);
    return;
  }
  launch_block_transform(x, ldx, T, C, B, H, W, C, 8, at.tw, 1, 0.f, 1.f, s);
  IGemm a = g; a.A = T; a.lda = C; a.epi_act = ACT_RELU; a.out = tmp; a.ldo = C;
  conv(at.q0, a, s);
  IGemm b = g; b.A = tmp; b.lda = C; b.epi_act = ACT_SIGMOID; b.mul = T; b.ldmul = C; b.rowscale = rowscale;
  b.res = res; b.ldres = ldres; b.out = out; b.ldo = ldo;
  conv(at.q2, b, s);
}

// AVIF frequency block (avif_inference.py:222-256).  In: plan_.hB.  Out: plan_.S = x + xt*avg*color*edge.
void Context::freq_avif(const BlockW& bw, int H, int W, hipStream_t s) {
  Plan& p = plan_;
  const int B = fb_, C = bw.cout;
  avif_transform(bw.at, p.hB, C, C, H, W, p.xd, p.hid, p.xt, C, nullptr, nullptr, 0, s);
  const int scales[4] = {1, 2, 4, 8};
  const bool one_pass = pool8_supported(H, W, C);   // all four scales from one read of x
  if (one_pass) launch_pool8_sums(p.hB, C, p.pool8, B, H, W, C, s);
  // the four pooled gate maps (avif_inference.py:227-234): each scale has its own two 1x1 layers on an S x S map -- tiny,
  // latency-bound GEMMs, so the four first layers go out as ONE grouped launch and the four second layers as another
  IGemm l1[4], l2[4];
  size_t goff = 0, poff = 0, hoff = 0;
  for (int i = 0; i < 4; ++i) {
    const int S = scales[i];
    float* pooled = p.pool + poff;
    float* hidden = p.pg1 + hoff;
    if (one_pass) launch_pool_from8(p.pool8, pooled, C, B, H, W, C, S, s);
    else launch_adaptive_avgpool(p.hB, C, pooled, C, B, H, W, C, S, s);
    IGemm g; g.B = B; g.H = S; g.W = S;
    IGemm a = g; a.A = pooled; a.lda = C; a.epi_act = ACT_RELU; a.out = hidden; a.ldo = C / 4;
    const ConvW& w1 = bw.ms1[i];
    a.Wt = w1.w; a.Wt16 = w1.w16; a.bias = w1.b; a.Cout = w1.cout; a.Cin = w1.cinp; a.taps = w1.taps;
    l1[i] = a;
    // the gate maps stay at S x S; F.interpolate(size=x.shape[-2:]) + the average happen inside avif_combine_k
    IGemm b = g; b.A = hidden; b.lda = C / 4; b.epi_act = ACT_SIGMOID; b.out = p.pg2 + goff; b.ldo = C;
    const ConvW& w3 = bw.ms3[i];
    b.Wt = w3.w; b.Wt16 = w3.w16; b.bias = w3.b; b.Cout = w3.cout; b.Cin = w3.cinp; b.taps = w3.taps;
    l2[i] = b;
    goff += (size_t)B * S * S * C;
    poff += (size_t)B * S * S * C;
    hoff += (size_t)B * S * S * (C / 4);
  }
  launch_igemm_group(l1, 4, s);
  launch_igemm_group(l2, 4, s);
  IGemm g; g.B = B; g.H = H; g.W = W;
  if (use_gemm16()) {   // colour + edge gates on the bf16 copy of x (plan_.hB16, refreshed by out_proj)
    IGemm c0 = g; c0.A16 = p.hB16; c0.lda16 = C; c0.epi_act = ACT_RELU; c0.out16 = p.hid16; c0.ldo16 = C;
    conv16(bw.cc0, c0, s);
    IGemm c2 = g; c2.A16 = p.hid16; c2.lda16 = C; c2.epi_act = ACT_SIGMOID; c2.rowscale = p.boost; c2.out = p.col; c2.ldo = C;
    conv16(bw.cc2, c2, s);
    IGemm e0 = g; e0.A16 = p.hB16; e0.lda16 = C; e0.epi_act = ACT_RELU; e0.out16 = p.hid16; e0.ldo16 = C / 2;
    conv16(bw.ep0, e0, s);
    IGemm e2 = g; e2.A16 = p.hid16; e2.lda16 = C / 2; e2.epi_act = ACT_SIGMOID; e2.rowscale = p.boost2; e2.out = p.edge; e2.ldo = C;
    conv16(bw.ep2, e2, s);
    launch_avif_combine(p.hB, p.xt, p.pg2, p.col, p.edge, p.S, B, H, W, C, s);
    return;
  }
  IGemm c0 = g; c0.A = p.hB; c0.lda = C; c0.epi_act = ACT_RELU; c0.out = p.hid; c0.ldo = C;
  conv(bw.cc0, c0, s);
  IGemm c2 = g; c2.A = p.hid; c2.lda = C; c2.epi_act = ACT_SIGMOID; c2.rowscale = p.boost; c2.out = p.col; c2.ldo = C;
  conv(bw.cc2, c2, s);
  IGemm e0 = g; e0.A = p.hB; e0.lda = C; e0.epi_act = ACT_RELU; e0.out = p.hid; e0.ldo = C / 2;
  conv(bw.ep0, e0, s);
  IGemm e2 = g; e2.A = p.hid; e2.lda = C / 2; e2.epi_act = ACT_SIGMOID; e2.rowscale = p.boost2; e2.out = p.edge; e2.ldo = C;
  conv(bw.ep2, e2, s);
  launch_avif_combine(p.hB, p.xt, p.pg2, p.col, p.edge, p.S, B, H, W, C, s);
}

void Context::conv16(const ConvW& cw, IGemm p, hipStream_t s, const float* bias_override) {
  p.Wt = nullptr; p.Wt16 = cw.w16; p.bias = bias_override ? bias_override : cw.b; p.Cout = cw.cout16; p.Cin = cw.cin16; p.taps = cw.taps;
  p.zero16 = plan_.zero16; p.splitk_ws = plan_.splitk_ws; p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
  p.tile256 = tile256_opt_; p.epi_direct_opt = epi_direct_opt_;
  if (!gemm16_supported(p)) throw std::logic_error("conv16: operand layout not supported by the DMA GEMM");
  // algorithmic FLOPs: 2 M N K over the real channel counts (a stacked gate pair counts its two layers, not the zero blocks)
  const double M = (double)p.B * p.H * p.W;
  prof_begin(PROF_GEMM, 2.0 * M * (p.sel_C ? 0.5 : 1.0) * ((double)cw.cout * cw.cin * cw.taps + (p.Cinb ? (double)cw.cout * p.Cinb : 0.0)), s);
  launch_gemm16(p, s);
  prof_end(s);
}

// GroupNorm-apply + activation fused into a full-resolution 3x3 conv (conv_tile.hip) when the shape and the epilogue allow it
bool Context::conv_tile_try(const ConvW& cw, IGemm p, const void* x, int ldx, bool in16, const float* scale, const float* shift, int act,
                            void* raw16, int ldraw, hipStream_t s) {
  if (!conv_tile_opt_ || !cw.w16 || cw.is_f8) return false;
  // measured (tools/option_ab.py conv_tile, profiles/r04_conv_tile_ab.txt): -0.09 ... -0.16 ms of a 512 x 512, B = 8 timestep
  // (2.1 M pixels), +0.07 ms at 256 x 256, B = 16 (1.05 M): the persistent kernel's weight prologue and its GELU-bound loaders
  // need the longer tile walk.  1 = that policy, 2 = wherever supported.
  if (conv_tile_opt_ == 1 && (int64_t)p.B * p.H * p.W < 1500000) return false;
  p.Wt = nullptr; p.Wt16 = cw.w16; p.bias = cw.b; p.Cout = cw.cout16; p.Cin = cw.cin16; p.taps = cw.taps;
  p.A16 = nullptr;
  if (!conv_tile_supported(p)) return false;
  prof_begin(PROF_GEMM, 2.0 * (double)p.B * p.H * p.W * (double)cw.cout * cw.cin * cw.taps, s);
  launch_conv_tile(p, x, ldx, in16, scale, shift, act, raw16, ldraw, s);
  prof_end(s);
  return true;
}

void Context::conv8(const ConvW& cw, IGemm p, hipStream_t s, const float* bias_override) {
  p.Wt = nullptr; p.Wt16 = cw.w8; p.bias = bias_override ? bias_override : cw.b; p.Cout = cw.cout16; p.Cin = cw.cin16; p.taps = cw.taps;
  p.f8 = 1; p.acc_scale = cw.w8_inv;
  p.zero16 = plan_.zero16; p.splitk_ws = plan_.splitk_ws; p.splitk_ws_bytes = GEMM16_SPLITK_WS_BYTES;
  p.tile256 = tile256_opt_; p.epi_direct_opt = epi_direct_opt_;
  if (!gemm16_supported(p)) throw std::logic_error("conv8: operand layout not supported by the fp8 GEMM");
  prof_begin(PROF_GEMM8, 2.0 * (double)p.B * p.H * p.W * ((double)cw.cout * cw.cin * cw.taps + (p.Cinb ? (double)cw.cout * p.Cinb : 0.0)), s);
  launch_gemm16(p, s);
  prof_end(s);
}

// bf16 mode, DCT families: every GEMM input is a bf16 tensor written once by its producer (GroupNorm-apply pass,
// previous GEMM epilogue, attention, DCT kernel), so both GEMM operands stream by LDS-DMA (gemm16.hip).
// The fp32 tensors that carry the residual stream and the GroupNorm statistics stay fp32.
bool Context::stream16() const {
  if (!use_gemm16() || cfg.arch == DDPMIR_ARCH_AVIF) return false;
  return stream16_opt_ < 0 ? cfg.dtype == DDPMIR_FP8 : stream16_opt_ != 0;
}

void Context::run_block16(const BlockW& bw, ActBuf in, int H, int W, ActBuf dest, hipStream_t s, double* out_stats, InStats st) {
  Plan& p = plan_;
  const int B = fb_, C = bw.cout, HW = H * W;
  const int cin8 = round_up(bw.cin, 8);
  const int ldin = in.ld;
  const bool in16 = in.h;   // bf16 activation stream: the input is read as bf16 and doubles as the 1x1 shortcut's operand
#define STAGE(what) debug_stage(s, bw.name.c_str(), what, B, H, W, C)
  IGemm g; g.B = B; g.H = H; g.W = W; g.bs = k.dct_block; g.low = k.low_size;
  // norm1 -> conv1 (+ bias + time projection)
  // statistics of (a channel range of) the input: the sums the producing max-pool / bilinear pass left where there are any,
  // the skip half's sums from the down block's conv_out epilogue, else a pass over the tensor
  auto norm1_part = [&](int c0, int C1, int G, const double* sums, int splits, int ld, int zero_to) {
    const float *ga = bw.n1g + c0, *be = bw.n1b + c0;
    float *sc = p.sc1 + c0, *sh = p.sh1 + c0;
    if (sums) { launch_gn_from_sums(sums, B, HW, C1, G, ga, be, sc, sh, s, ld, zero_to); return; }
    if (!splits && in16) splits = launch_stats16(in.at(c0).p, ldin, B, H, W, C1, G, p.gn_partial, s);
    if (splits) launch_gn_from_partials(p.gn_partial, splits, B, (double)HW * (C1 / G), C1, G, ga, be, sc, sh, s, ld, zero_to);
    else launch_gn_scale_shift(in.f() + c0, ldin, B, HW, C1, G, ga, be, p.gn_partial, sc, sh, s, ld, zero_to);
  };
  if (st.concat) {
    // up block: the input is cat[upsampled, skip] and its 8 groups split 4 + 4 over the halves
    const int ch = bw.cin / 2, ld = round_up(bw.cin, 4);
    norm1_part(0, ch, 4, nullptr, st.partial_splits, ld, 0);
    norm1_part(ch, ch, 4, st.skip_sums, 0, ld, 0);
  } else {
    norm1_part(0, bw.cin, gn_groups(bw.cin), nullptr, st.partial_splits, 0, -1);
  }
  STAGE("norm1 statistics");
  // (3-channel image input: x4 carries a zero 4th channel, gn_finalize zeroes its scale/shift, a16 pads to 8)
  IGemm c1 = g; c1.A16 = p.a16; c1.lda16 = cin8; c1.tb = p.tproj + bw.tproj_off; c1.ldtb = nt_total_;
  c1.out = p.hA; c1.ldo = C;
  // norm2's statistics come out of conv1's epilogue where the launch is not split-K (levels 0-3 at 256x256); the
  // accumulator is cleared by the norm1-apply pass that runs just before
  const bool no_gn_fuse = exp_switch("DDPMIR_NO_GN_FUSE");
  IGemm probe = c1; probe.Cout = bw.conv1.cout16; probe.Cin = bw.conv1.cin16; probe.taps = bw.conv1.taps;
  const bool fuse_gn = !no_gn_fuse && gemm16_can_fuse_gn(probe, gn_groups(C));
  // (DDPMIR_FP8: the deep 3x3 convs take their operand as e4m3 instead -- written by the same pass)
  const bool c1_f8 = bw.conv1.w8 != nullptr, c2_f8 = bw.conv2.w8 != nullptr, co_f8 = bw.conv_out.w8 != nullptr;
  launch_gn_apply16(in16 ? nullptr : in.f(), ldin, B, HW, bw.cinp, p.sc1, p.sh1, ACT_NONE, c1_f8 ? nullptr : p.a16, cin8,
                    bw.has_shortcut && !in16 ? p.x16 : nullptr, cin8, s, nullptr, 0, c1_f8 ? p.a8 : nullptr, bw.conv1.cin16,
                    in16 ? in.p : nullptr);
  const void* xs16 = in16 ? in.p : p.x16;          // raw bf16 block input: operand of the 1x1 shortcut
  const int ldxs = in16 ? ldin : cin8;
  STAGE("norm1 apply");
  // With the statistics taken from the fp32 accumulators, conv1's output is only ever read by the norm2-apply pass: it is
  // written once as bf16 (half the bytes out and in again); the deep levels (split-K, statistics pass over the tensor) keep fp32
  const bool ha16 = fuse_gn && !exp_switch("DDPMIR_HA_FP32");
  if (fuse_gn) { c1.gn_sums = p.gn_sums; c1.gn_groups = gn_groups(C); }
  if (ha16) { c1.out = nullptr; c1.out16 = p.hid16; c1.ldo16 = C; }
  if (c1_f8) { c1.A16 = p.a8; c1.lda16 = bw.conv1.cin16; conv8(bw.conv1, c1, s); }
  else conv16(bw.conv1, c1, s);
  STAGE("conv1");
  // norm2 -> GELU -> conv2 ; hB also as bf16 for the qkv projection
  if (fuse_gn) launch_gn_from_sums(p.gn_sums, B, HW, C, gn_groups(C), bw.n2g, bw.n2b, p.sc2, p.sh2, s);
  else launch_gn_scale_shift(p.hA, C, B, HW, C, gn_groups(C), bw.n2g, bw.n2b, p.gn_partial, p.sc2, p.sh2, s);
  STAGE("norm2 statistics");
  const bool attn = bw.level >= cfg.attn_min_level;
  // 64- and 128-channel blocks of the DCT families: DCT, both gate layers and S = x + g * xd run as one pass over x
  const bool fuse_freq = cfg.arch != DDPMIR_ARCH_AVIF && !exp_switch("DDPMIR_SPLIT_GATES") && !exp_switch("DDPMIR_NO_FREQ_FUSE") &&
                         freq_gate_fused_supported(B, H, W, C, k.dct_block);   // (e4m3 conv_out: the kernel writes S as e4m3)
  // conv2's output x: where the one-pass frequency kernel is its only fp32 reader (no attention, or the 64-channel blocks whose
  // out_proj + residual is formed inside that kernel) x is written as bf16 alone and read back as bf16
  const bool hb16_only = fuse_freq && (!attn || C == 64) && !exp_switch("DDPMIR_HB_FP32");
  IGemm c2 = g; c2.A16 = p.a16; c2.lda16 = C; c2.out = hb16_only ? nullptr : p.hB; c2.ldo = C;
  if (attn || hb16_only) { c2.out16 = p.hB16; c2.ldo16 = C; }
  // full-resolution blocks: norm2 + GELU are applied inside the conv (conv_tile.hip) -- no operand pass, the raw tensor is read once
  if (!c2_f8 && conv_tile_try(bw.conv2, c2, ha16 ? (const void*)p.hid16 : (const void*)p.hA, C, ha16, p.sc2, p.sh2, ACT_GELU, nullptr, 0, s)) {
    STAGE("norm2 + GELU + conv2 (one kernel)");
  } else {
    launch_gn_apply16(p.hA, C, B, HW, C, p.sc2, p.sh2, ACT_GELU, c2_f8 ? nullptr : p.a16, C, nullptr, 0, s, nullptr, 0,
                      c2_f8 ? p.a8 : nullptr, C, ha16 ? p.hid16 : nullptr);
    STAGE("norm2 apply + GELU");
    if (c2_f8) { c2.A16 = p.a8; conv8(bw.conv2, c2, s); }
    else conv16(bw.conv2, c2, s);
    STAGE("conv2");
  }
  if (attn) {
    // head_dim 16 levels: the projection's epilogue writes the attention operand images itself (no fp32 qkv round trip)
    const bool no_fuse = exp_switch("DDPMIR_NO_QKV_FUSE");
    const bool prepack = !no_fuse && attention_bf16_prepack_supported(HW, C, k.heads);
    IGemm q = g; q.A16 = p.hB16; q.lda16 = C;
    if (prepack) {
      const AttnPack16 t = attention_bf16_pack_targets(B, HW, C, k.heads, p.attn_ws);
      q.pk_q = t.q; q.pk_kv = t.kv; q.pk_kmax = t.kmax; q.pk_nt = t.ntiles; q.pk_qscale = t.qscale; q.pk_D = t.D; q.pk_heads = t.heads; q.pk_qmax = t.qmax2;
    } else {
      q.out = p.qkv; q.ldo = 3 * C;
    }
    conv16(bw.in_proj, q, s);
  STAGE("attention in_proj");
    const double nn = (double)HW;
    prof_begin(bw.level == 0 ? PROF_ATTN_L0 : PROF_ATTN, 4.0 * nn * nn * C * B, s);
    if (attention_bf16_supported(HW, C, k.heads)) {
      const AttnOpts ao = attn_opts();
      // two concurrent lanes (loop_two_lanes): the full-resolution attention launches of the two lanes take turns -- each runs
      // beside the OTHER lane's convolutions / streaming passes / deep levels, never beside the other lane's attention
      if (bw.level == 0 && attn_wait_) DDPMIR_HIP_CHECK(hipStreamWaitEvent(s, attn_wait_, 0));
      launch_attention_bf16(prepack ? nullptr : p.qkv, p.att, C, B, HW, C, k.heads, p.attn_ws, s, p.att16, &ao);
      if (bw.level == 0 && attn_done_) DDPMIR_HIP_CHECK(hipEventRecord(attn_done_, s));
    } else {
      launch_attention(p.qkv, p.att, C, B, HW, C, k.heads, s);
      launch_gn_apply16(p.att, C, B, HW, C, nullptr, nullptr, ACT_NONE, p.att16, C, nullptr, 0, s);
    }
    prof_end(s);
    STAGE("attention core");
    // 64-channel blocks: the output projection and its residual are formed inside the frequency-block kernel (level 0:
    // 135 + 165 us -> 231 us); at 128 channels the extra GEMM on 64-pixel tiles costs more than the pass it saves (176 vs 150 us)
    if (!(fuse_freq && C == 64)) {
      IGemm o = g; o.A16 = p.att16; o.lda16 = C; o.res = p.hB; o.ldres = C; o.out = p.hB; o.ldo = C;
      if (cfg.arch == DDPMIR_ARCH_AVIF) { o.out16 = p.hB16; o.ldo16 = C; }   // the colour / edge gates read x as bf16
      conv16(bw.out_proj, o, s);
      STAGE("attention out_proj");
    }
  } else if (cfg.arch == DDPMIR_ARCH_AVIF) {
    launch_gn_apply16(p.hB, C, B, HW, C, nullptr, nullptr, ACT_NONE, p.hB16, C, nullptr, 0, s);
  }
  // frequency block: DCT -> masked bf16 operands -> gates -> S (bf16 only, it feeds conv_out alone)
  const bool split_gates = exp_switch("DDPMIR_SPLIT_GATES");   // A/B: the four separate gate GEMMs
  if (cfg.arch == DDPMIR_ARCH_AVIF) {
    freq_avif(bw, H, W, s);                                            // -> plan_.S (fp32)
    launch_gn_apply16(p.S, C, B, HW, C, nullptr, nullptr, ACT_NONE, co_f8 ? nullptr : p.S16, C, nullptr, 0, s, nullptr, 0,
                      co_f8 ? p.S8 : nullptr, C);
  } else if (fuse_freq) {
    // (freq_fused.hip; with attention in the block, x = out_proj(att) + x is formed there too)
    launch_freq_gate_fused(hb16_only ? nullptr : p.hB, bw.dct, bw.fg1.w16, bw.fg1.b, bw.fg2.w16, bw.fg2.b, p.boost, co_f8 ? nullptr : p.S16, B, H, W, C,
                           k.dct_block, k.low_size, s, attn && C == 64 ? p.att16 : nullptr, bw.out_proj.w16, bw.out_proj.b,
                           co_f8 ? p.S8 : nullptr, hb16_only ? p.hB16 : nullptr);
  } else if (!split_gates) {
    launch_block_transform(p.hB, C, p.xd, C, B, H, W, C, k.dct_block, bw.dct, 0, 0.f, 1.f, s, p.xdl16, nullptr, k.low_size);
    IGemm a = g; a.A16 = p.xdl16; a.lda16 = C; a.epi_act = ACT_LRELU; a.out16 = p.hid16; a.ldo16 = C;
    conv16(bw.fg1, a, s);
    IGemm d = g; d.A16 = p.hid16; d.lda16 = C; d.epi_act = ACT_SIGMOID; d.mul = p.xd; d.ldmul = C; d.sel_C = C;
    d.rowscale = p.boost; d.res = p.hB; d.ldres = C;
    if (co_f8) { d.out8 = p.S8; d.ldo8 = C; } else { d.out16 = p.S16; d.ldo16 = C; }
    conv16(bw.fg2, d, s);
  } else {
    launch_block_transform(p.hB, C, p.xd, C, B, H, W, C, k.dct_block, bw.dct, 0, 0.f, 1.f, s, p.xdl16, p.xdh16, k.low_size);
    IGemm a = g; a.A16 = p.xdl16; a.lda16 = C; a.epi_act = ACT_LRELU; a.out16 = p.hid16; a.ldo16 = C / 2;
    conv16(bw.low0, a, s);
    IGemm b2 = g; b2.A16 = p.hid16; b2.lda16 = C / 2; b2.epi_act = ACT_SIGMOID; b2.mul = p.xd; b2.ldmul = C;
    b2.epi_mask = MASK_LOW; b2.res = p.hB; b2.ldres = C; b2.out = p.S; b2.ldo = C;
    conv16(bw.low2, b2, s);
    IGemm c = a; c.A16 = p.xdh16;
    conv16(bw.high0, c, s);
    IGemm d = b2; d.epi_mask = MASK_HIGH; d.rowscale = p.boost; d.res = p.S; d.ldres = C; d.out = nullptr;
    if (co_f8) { d.out8 = p.S8; d.ldo8 = C; } else { d.out16 = p.S16; d.ldo16 = C; }
    conv16(bw.high2, d, s);
  }
  STAGE("frequency block");
  // conv_out + residual; a 1x1 shortcut conv rides along as a second K segment of the same GEMM (identity shortcuts
  // are the plain fp32 residual)
  IGemm co = g; co.A16 = p.S16; co.lda16 = C;
  if (dest.h) { co.out16 = dest.p; co.ldo16 = dest.ld; } else { co.out = dest.f(); co.ldo = dest.ld; }
  auto identity_residual = [&](IGemm& q) { if (in16) { q.res16 = in.p; q.ldres = ldin; } else { q.res = in.f(); q.ldres = ldin; } };
  if (out_stats) { co.gn_sums = out_stats; co.gn_groups = 4; }   // a down block's output is the skip half of an up block's norm1
  const bool split_sc = exp_switch("DDPMIR_SPLIT_SHORTCUT");   // A/B: separate shortcut GEMM
  if (co_f8) {
    // e4m3 conv_out: the bf16 1x1 shortcut still rides along as the second K segment -- its stages run on the bf16 MFMA into the
    // same accumulators, with weights pre-multiplied by conv_out's power-of-two fp8 weight scale (BlockW::sc16s)
    co.A16 = p.S8;
    if (bw.has_shortcut) {
      co.A16b = xs16; co.lda16b = ldxs; co.Wt16b = bw.sc16s; co.Cinb = bw.shortcut.cin16;
      conv8(bw.conv_out, co, s, bw.co_bias);
    } else {
      identity_residual(co);
      conv8(bw.conv_out, co, s);
    }
  } else if (bw.has_shortcut && !split_sc) {
    co.A16b = xs16; co.lda16b = ldxs; co.Wt16b = bw.shortcut.w16; co.Cinb = bw.shortcut.cin16;
    conv16(bw.conv_out, co, s, bw.co_bias);
  } else {
    if (bw.has_shortcut) {
      IGemm sc = g; sc.A16 = xs16; sc.lda16 = ldxs;
      if (dest.h) { sc.out16 = dest.p; sc.ldo16 = dest.ld; co.res16 = dest.p; } else { sc.out = dest.f(); sc.ldo = dest.ld; co.res = dest.f(); }
      conv16(bw.shortcut, sc, s);
      co.ldres = dest.ld;
    } else {
      identity_residual(co);
    }
    conv16(bw.conv_out, co, s);
  }
  STAGE("conv_out (+ shortcut)");
#undef STAGE
}

// Can the down block at `level` emit the GroupNorm sums of its output (the skip half of the matching up block's norm1 input)
// from conv_out's epilogue?  Decided from shapes alone, so the down and the up block agree.
bool Context::skip_stats_fused(int level, int B, int H, int W) const {
  if (!use_gemm16() || level > 4 || exp_switch("DDPMIR_NO_SKIP_STATS")) return false;
  const BlockW& bw = blocks_[level];
  IGemm probe; probe.B = B; probe.H = H; probe.W = W; probe.Cout = bw.conv_out.cout16; probe.Cin = bw.conv_out.cin16; probe.taps = 9;
  probe.out = plan_.hA;   // (any non-null output)
  return gemm16_can_fuse_gn(probe, 4);
}

void Context::run_block(const BlockW& bw, ActBuf in_a, int H, int W, ActBuf dest_a, hipStream_t s, double* out_stats, InStats st) {
  if (use_gemm16()) { run_block16(bw, in_a, H, W, dest_a, s, out_stats, st); return; }
  const float* in = in_a.f(); const int ldin = in_a.ld;
  float* dest = dest_a.f(); const int lddest = dest_a.ld;
  Plan& p = plan_;
  const int B = fb_, C = bw.cout;
  IGemm g; g.B = B; g.H = H; g.W = W;
  // norm1 -> conv1 (+ bias + time projection)            webp_inference.py:304-309
  launch_gn_scale_shift(in, ldin, B, H * W, bw.cin, gn_groups(bw.cin), bw.n1g, bw.n1b, p.gn_partial, p.sc1, p.sh1, s);
  IGemm c1 = g; c1.A = in; c1.lda = ldin; c1.scale = p.sc1; c1.shift = p.sh1;
  c1.tb = p.tproj + bw.tproj_off; c1.ldtb = nt_total_; c1.out = p.hA; c1.ldo = C;
  conv(bw.conv1, c1, s);
  // norm2 -> GELU -> conv2                               webp_inference.py:311-314
  launch_gn_scale_shift(p.hA, C, B, H * W, C, gn_groups(C), bw.n2g, bw.n2b, p.gn_partial, p.sc2, p.sh2, s);
  IGemm c2 = g; c2.A = p.hA; c2.lda = C; c2.scale = p.sc2; c2.shift = p.sh2; c2.pro_act = ACT_GELU;
  c2.out = p.hB; c2.ldo = C;
  conv(bw.conv2, c2, s);
  // self-attention over all H*W tokens + residual        webp_inference.py:317-321
  if (bw.level >= cfg.attn_min_level) {
    IGemm q = g; q.A = p.hB; q.lda = C; q.out = p.qkv; q.ldo = 3 * C;
    conv(bw.in_proj, q, s);
    // algorithmic FLOPs of QK^T + PV: 4 * N^2 * C per image
    const double nn = (double)H * W;
    prof_begin(bw.level == 0 ? PROF_ATTN_L0 : PROF_ATTN, 4.0 * nn * nn * C * B, s);
    if (use_gemm16() && attention_bf16_supported(H * W, C, k.heads))
      launch_attention_bf16(p.qkv, p.att, C, B, H * W, C, k.heads, p.attn_ws, s);
    else
      launch_attention(p.qkv, p.att, C, B, H * W, C, k.heads, s);
    prof_end(s);
    IGemm o = g; o.A = p.att; o.lda = C; o.res = p.hB; o.ldres = C; o.out = p.hB; o.ldo = C;
    conv(bw.out_proj, o, s);
  }
  // frequency block -> S                                  webp_inference.py:324
  if (cfg.arch == DDPMIR_ARCH_AVIF) freq_avif(bw, H, W, s);
  else freq_dct(bw, H, W, s);
  // shortcut + conv_out                                   webp_inference.py:270,327
  const float* res = in; int ldres = ldin;
  if (bw.has_shortcut) {
    IGemm sc = g; sc.A = in; sc.lda = ldin; sc.out = dest; sc.ldo = lddest;
    conv(bw.shortcut, sc, s);
    res = dest; ldres = lddest;
  }
  IGemm co = g; co.A = p.S; co.lda = C; co.res = res; co.ldres = ldres; co.out = dest; co.ldo = lddest;
  conv(bw.conv_out, co, s);
}

// The U-Net forward in three pieces so that the two-lane sampling loop can run the full-resolution blocks per lane (their
// inputs / outputs are what the host codec produces / consumes) and everything below level 0 ONCE for both lanes:
//   fwd_head(rows)  image -> NHWC, time embedding, down1                      for samples [row0, row0 + nb)
//   fwd_mid(B)      max-pool, down2 .. bottleneck .. up4                      for all B samples
//   fwd_tail(rows)  up5, head conv, tanh -> NCHW                              for samples [row0, row0 + nb)
// forward() = the three pieces on the whole batch (same launch sequence as one function).
namespace {
// per-sample side inputs (time projections, cl boosts) of a row range: the block code indexes them from sample 0
struct SideRows {
  float *&tproj, *&boost, *&boost2, *&constmix;
  float *t0, *b0, *b20, *c0;
  SideRows(float*& tp, float*& bo, float*& bo2, float*& cm, int row0, int nt)
      : tproj(tp), boost(bo), boost2(bo2), constmix(cm), t0(tp), b0(bo), b20(bo2), c0(cm) {
    tproj += (size_t)row0 * nt; boost += row0; boost2 += row0; constmix += row0;
  }
  ~SideRows() { tproj = t0; boost = b0; boost2 = b20; constmix = c0; }
};
}  // namespace

void Context::prof_close_forward(hipStream_t s) {
  if (!prof_on_) return;   // close the "forward" span (the attention spans nested inside are already closed)
  for (auto it = prof_spans_.rbegin(); it != prof_spans_.rend(); ++it)
    if (it->cls == PROF_FWD) { DDPMIR_HIP_CHECK(hipEventRecord(it->b, s)); break; }
}

// bilinear x2 of `in` (C channels) into the first half of the concat buffer `cat`; in bf16 / fp8 mode the pass also leaves the
// GroupNorm(4 groups) sums of what it wrote in plan_.gn_partial: returns their splits per sample (0: none)
int Context::upsample(ActBuf in, int C, ActBuf cat, int B, int Hi, int Wi, hipStream_t s) {
  if (in.h) return launch_upsample2_16(in.p, C, cat.p, cat.ld, B, Hi, Wi, C, s, plan_.gn_partial, 4);
  if (use_gemm16() && stream_stats_supported(C, 4, 4) && !exp_switch("DDPMIR_NO_STREAM_STATS"))
    return launch_upsample2_stats(in.f(), C, cat.f(), cat.ld, B, Hi, Wi, C, s, plan_.gn_partial, 4);
  launch_upsample2(in.f(), C, cat.f(), cat.ld, B, Hi, Wi, C, s);
  return 0;
}

void Context::fwd_head(const float* x, const float* t, const float* cl, int row0, int nb, bool whole, hipStream_t s) {
  Plan& p = plan_;
  const int H = p.H, W = p.W;
  const size_t HW = (size_t)H * W;
  fb_ = nb;
  prof_begin(PROF_FWD, 0.0, s);
  if (!whole && prof_on_ && !prof_spans_.empty()) prof_spans_.back().count = 0;   // a piece: its time counts, not its number
  float* x4 = p.x4 + (size_t)row0 * HW * 4;
  launch_nchw3_to_nhwc4(x, x4, nb, H, W, s);
  debug_stage(s, "forward", "image layout NCHW3 -> NHWC4", nb, H, W, 3);
  launch_time_embed(t, nb, tw_, p.hidden + (size_t)row0 * 1024, p.temb + (size_t)row0 * 256, p.tproj + (size_t)row0 * nt_total_, s);
  if (cfg.arch == DDPMIR_ARCH_AVIF) {
    launch_cl_boost(cl, nb, 0.5f, 0.5f, 0.3f, 1.5f, p.boost + row0, s);    // colour   avif_inference.py:246
    launch_cl_boost(cl, nb, 0.7f, 0.3f, 0.5f, 1.3f, p.boost2 + row0, s);   // edge     avif_inference.py:247
  } else {
    launch_cl_boost(cl, nb, 0.f, 1.f, k.boost_lo, k.boost_hi, p.boost + row0, s);   // webp_inference.py:263
  }
  {
    SideRows rows(p.tproj, p.boost, p.boost2, p.constmix, row0, nt_total_);
    // the skip tensor goes straight into the second half of up5's concat buffer
    debug_stage(s, "forward", "image layout + time embedding", nb, H, W, 3);
    double* st0 = skip_stats_fused(0, nb, H, W) ? p.skip_sums[0] + gn_sums_doubles(row0, (int)HW, 4) : nullptr;
    ActBuf img; img.p = x4; img.ld = 4;
    run_block(blocks_[0], img, H, W, p.cat[0].at((size_t)row0 * HW * 128 + 64), s, st0);
  }
  if (!whole) prof_close_forward(s);
}

void Context::fwd_mid(int B, bool whole, hipStream_t s) {
  Plan& p = plan_;
  const int H = p.H, W = p.W;
  fb_ = B;
  if (!whole) { prof_begin(PROF_FWD, 0.0, s); if (prof_on_ && !prof_spans_.empty()) prof_spans_.back().count = 0; }
  auto hh = [&](int l) { return H >> l; };
  auto ww = [&](int l) { return W >> l; };
  // skip tensors are written straight into the second half of the matching up block's concat buffer
  const int cat_c[5] = {128, 256, 512, 1024, 1024};
  const int skip_off[5] = {64, 128, 256, 512, 512};
  const int skip_c[5] = {64, 128, 256, 512, 512};
  // down path
  // (bf16 stream: the pool / bilinear pass also leaves the GroupNorm sums of what it wrote -- the next block's norm1 input)
  auto pool = [&](int l) {   // level l's skip tensor -> pin[l + 1]; returns the splits of the sums left in p.gn_partial
    const ActBuf src = p.cat[l].at(skip_off[l]);
    int splits = 0;
    const int G = gn_groups(skip_c[l]);
    if (src.h) splits = launch_maxpool2_16(src.p, cat_c[l], p.pin[l + 1].p, skip_c[l], B, hh(l), ww(l), skip_c[l], s, p.gn_partial, G);
    else if (use_gemm16() && stream_stats_supported(skip_c[l], G, 4) && !exp_switch("DDPMIR_NO_STREAM_STATS"))
      splits = launch_maxpool2_stats(src.f(), cat_c[l], p.pin[l + 1].f(), skip_c[l], B, hh(l), ww(l), skip_c[l], s, p.gn_partial, G);
    else launch_maxpool2(src.f(), cat_c[l], p.pin[l + 1].f(), skip_c[l], B, hh(l), ww(l), skip_c[l], s);
    debug_stage(s, "forward", "max-pool", B, hh(l), ww(l), skip_c[l]);
    return splits;
  };
  InStats st; st.partial_splits = pool(0);
  for (int l = 1; l < 5; ++l) {
    run_block(blocks_[l], p.pin[l], hh(l), ww(l), p.cat[l].at(skip_off[l]), s,
              skip_stats_fused(l, B, hh(l), ww(l)) ? p.skip_sums[l] : nullptr, st);
    st.partial_splits = pool(l);
  }
  run_block(blocks_[5], p.pin[5], hh(5), ww(5), p.bb0, s, nullptr, st);
  run_block(blocks_[6], p.bb0, hh(5), ww(5), p.bb1, s);
  run_block(blocks_[7], p.bb1, hh(5), ww(5), p.bb2, s);
  // up path: cat[upsampled, skip]                           webp_inference.py:389-393
  ActBuf prev = p.bb2; int prev_c = 512;
  for (int l = 4; l >= 1; --l) {
    InStats us;
    us.skip_sums = skip_stats_fused(l, B, hh(l), ww(l)) ? p.skip_sums[l] : nullptr;
    us.partial_splits = upsample(prev, prev_c, p.cat[l], B, hh(l + 1), ww(l + 1), s);
    us.concat = us.partial_splits || us.skip_sums != nullptr;
    debug_stage(s, "forward", "bilinear x2", B, hh(l + 1), ww(l + 1), prev_c);
    const BlockW& bw = blocks_[8 + (4 - l)];
    run_block(bw, p.cat[l], hh(l), ww(l), p.u[l], s, nullptr, us);
    prev = p.u[l]; prev_c = bw.cout;
  }
  if (!whole) prof_close_forward(s);
}

void Context::fwd_tail(float* out, int row0, int nb, bool whole, hipStream_t s) {
  Plan& p = plan_;
  const int H = p.H, W = p.W;
  const size_t HW = (size_t)H * W;
  fb_ = nb;
  if (!whole) prof_begin(PROF_FWD, 0.0, s);            // (counts as the lane's forward)
  SideRows rows(p.tproj, p.boost, p.boost2, p.constmix, row0, nt_total_);
  const BlockW& bw = blocks_[12];
  const int prev_c = blocks_[11].cout;
  const ActBuf cat0 = p.cat[0].at((size_t)row0 * HW * 128);
  const ActBuf u0a = p.u[0].at((size_t)row0 * HW * bw.cout);
  const ActBuf u1 = p.u[1].at((size_t)row0 * (HW / 4) * prev_c);
  InStats us;
  us.skip_sums = skip_stats_fused(0, nb, H, W) ? p.skip_sums[0] + gn_sums_doubles(row0, (int)HW, 4) : nullptr;
  us.partial_splits = upsample(u1, prev_c, cat0, nb, H >> 1, W >> 1, s);
  us.concat = us.partial_splits || us.skip_sums != nullptr;
  run_block(bw, cat0, H, W, u0a, s, nullptr, us);
  float* u0 = u0a.f();   // (fp32 stream; the bf16 stream reads u0a below)
  // head: u5 + mix * transform(u5) -> GroupNorm(8) -> SiLU -> conv 64->3 -> tanh     webp_inference.py:396-399
  const bool head_gn_fused = use_gemm16() && cfg.arch != DDPMIR_ARCH_AVIF && block_transform_gn_fusable(H, W, 64, k.dct_block) &&
                             !exp_switch("DDPMIR_NO_HEAD_GN_FUSE");
  if (cfg.arch == DDPMIR_ARCH_AVIF)
    avif_transform(head_at_, u0, 64, 64, H, W, p.xd, p.hid, p.S, 64, p.constmix, u0, 64, s);
  else if (u0a.h)   // bf16 stream: u5 in, the mixed tensor out as bf16 (its only reader is the GroupNorm-apply pass below)
    launch_block_transform(u0, 64, reinterpret_cast<float*>(p.S16), 64, nb, H, W, 64, k.dct_block, head_dct_, 0, 1.0f, k.head_mix, s,
                           nullptr, nullptr, 0, head_gn_fused ? p.gn_sums : nullptr, 1);
  else
    launch_block_transform(u0, 64, p.S, 64, nb, H, W, 64, k.dct_block, head_dct_, 0, 1.0f, k.head_mix, s, nullptr, nullptr, 0,
                           head_gn_fused ? p.gn_sums : nullptr);
  // (DCT families, bf16 / fp8 mode: the mix pass also emits the GroupNorm sums of its output)
  if (head_gn_fused)
    launch_gn_from_partials(p.gn_sums, (H / k.dct_block) * (W / k.dct_block), nb, (double)H * W * 8, 64, 8, head_g_, head_b_, p.sc1, p.sh1, s);
  else if (u0a.h)
    launch_gn_from_partials(p.gn_partial, launch_stats16(p.S16, 64, nb, H, W, 64, 8, p.gn_partial, s), nb, (double)H * W * 8, 64, 8,
                            head_g_, head_b_, p.sc1, p.sh1, s);
  else
    launch_gn_scale_shift(p.S, 64, nb, H * W, 64, 8, head_g_, head_b_, p.gn_partial, p.sc1, p.sh1, s);
  if (use_gemm16() && head_fused_opt_ && head_fused_supported(H, W) && head_conv_.w16 && head_conv_.cout16 == 4 && head_conv_.cin16 == 64) {
    // one pass: GroupNorm-apply + SiLU + conv 64 -> 3 + tanh + NCHW (head_fused.hip) instead of three launches
    prof_begin(PROF_GEMM, 2.0 * (double)nb * H * W * 3 * 64 * 9, s);
    launch_head_fused(u0a.h ? (const void*)p.S16 : (const void*)p.S, 64, u0a.h, p.sc1, p.sh1, head_conv_.w16, head_conv_.b, out, nb, H, W, s);
    prof_end(s);
  } else if (use_gemm16()) {   // SiLU(GN(.)) as a bf16 operand -> DMA GEMM with Cout padded to 4 -> tanh + NCHW in one pass
    float* x4 = p.x4 + (size_t)row0 * HW * 4;          // the lane's image rows are free again
    launch_gn_apply16(u0a.h ? nullptr : p.S, 64, nb, H * W, 64, p.sc1, p.sh1, ACT_SILU, p.a16, 64, nullptr, 0, s, nullptr, 0, nullptr, 0,
                      u0a.h ? p.S16 : nullptr);
    IGemm hc; hc.B = nb; hc.H = H; hc.W = W; hc.A16 = p.a16; hc.lda16 = 64; hc.out = x4; hc.ldo = 4;
    conv16(head_conv_, hc, s);
    launch_nhwc4_to_nchw3_tanh(x4, out, nb, H, W, s);
  } else {
    IGemm hc; hc.B = nb; hc.H = H; hc.W = W; hc.A = p.S; hc.lda = 64; hc.scale = p.sc1; hc.shift = p.sh1;
    hc.pro_act = ACT_SILU; hc.epi_act = ACT_TANH; hc.out = out; hc.ldo = 0; hc.store = STORE_NCHW;
    conv(head_conv_, hc, s);
  }
  debug_stage(s, "forward", "head (transform mix, GroupNorm, SiLU, conv 64->3, tanh)", nb, H, W, 64);
  prof_close_forward(s);
}

void Context::forward(const float* x, const float* t, const float* cl, float* out, hipStream_t s) {
  const int B = fb_;
  fwd_head(x, t, cl, 0, B, true, s);
  fwd_mid(B, true, s);
  fwd_tail(out, 0, B, true, s);
  fb_ = B;
}

// Every entry point that reads caller device buffers on the context's own stream calls this first: the caller produced them on
// ITS stream -- the default stream, since the C ABI passes none -- and stream_ is non-blocking (no implicit ordering against
// the default stream), so the order is made explicit with an event (a C caller that did not device-synchronise would race).
void Context::order_after_caller() {
  DDPMIR_HIP_CHECK(hipEventRecord(lane_ev_[0], nullptr));
  DDPMIR_HIP_CHECK(hipStreamWaitEvent(stream_, lane_ev_[0], 0));
}

void Context::predict(const float* x, const float* t, const float* cl, float* out, int B, int H, int W, hipStream_t s) {
  if (!finalized_) throw std::logic_error("predict before finalize_weights");
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  ensure_plan(B, H, W);
  fb_ = B;
  const bool own = (s == nullptr);
  hipStream_t st = own ? stream_ : s;
  if (own) order_after_caller();
  Plan& p = plan_;
  const size_t n_img = (size_t)B * 3 * H * W;
  if (cfg.use_graph && !prof_on_) {
    // fixed-address staging buffers so the captured graph is replayable for any caller pointers
    DDPMIR_HIP_CHECK(hipMemcpyAsync(p.g_x, x, n_img * 4, hipMemcpyDeviceToDevice, st));
    DDPMIR_HIP_CHECK(hipMemcpyAsync(p.g_t, t, B * 4, hipMemcpyDeviceToDevice, st));
    DDPMIR_HIP_CHECK(hipMemcpyAsync(p.g_cl, cl ? cl : t, B * 4, hipMemcpyDeviceToDevice, st));
    if (!p.gexec) {
      DDPMIR_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      try {
        forward(p.g_x, p.g_t, p.g_cl, p.g_out, st);
      } catch (...) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(st, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
      }
      DDPMIR_HIP_CHECK(hipStreamEndCapture(st, &p.graph));
      DDPMIR_HIP_CHECK(hipGraphInstantiate(&p.gexec, p.graph, nullptr, nullptr, 0));
    }
    DDPMIR_HIP_CHECK(hipGraphLaunch(p.gexec, st));
    DDPMIR_HIP_CHECK(hipMemcpyAsync(out, p.g_out, n_img * 4, hipMemcpyDeviceToDevice, st));
  } else {
    forward(x, t, cl ? cl : t, out, st);
  }
  if (own) DDPMIR_HIP_CHECK(hipStreamSynchronize(st));
}

// ------------------------------------------------------------------------------------------------ phase fix
// `off` = element offset of the image sub-batch inside the per-plan DFT buffers (micro-batch lanes of the loop)
void Context::phase_prepare(const float* y, int B, int H, int W, hipStream_t s, size_t off) {
  Plan& p = plan_;
  const int P = B * 3;
  // dft[0], dft[1] = unit phase of fft2(y), from a double-precision transform (see sampler_kernels.hip)
  DftPlan64 d{p.tw64_ch, p.tw64_sh, H, p.tw64_cw, p.tw64_sw, W};
  launch_ref_unit_phase_f64(y, p.dft[0] + off, p.dft[1] + off, p.dft64_re + off, p.dft64_im + off, P, d, s);
}

void Context::phase_fix(float* x, int B, int H, int W, double alpha, hipStream_t s, size_t off) {
  Plan& p = plan_;
  DftPlan d{p.tw_ch, p.tw_sh, H, p.tw_cw, p.tw_sw, W};
  const int P = B * 3;
  launch_dft2_forward(x, p.dft[2] + off, p.dft[3] + off, p.dft[4] + off, p.dft[5] + off, P, d, s);
  launch_phase_apply_inverse(x, p.dft[2] + off, p.dft[3] + off, p.dft[0] + off, p.dft[1] + off, p.dft[4] + off,
                             p.dft[5] + off, p.dft[6] + off, p.dft[7] + off, x, P, d, (float)alpha, (float)(1.0 - alpha), s);
}

void Context::phase_consistency(const float* x, const float* ref, float* out, int B, int H, int W, double alpha) {
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  ensure_plan(B, H, W);
  order_after_caller();
  Plan& p = plan_;
  const size_t n = (size_t)B * 3 * H * W;
  phase_prepare(ref, B, H, W, stream_, 0);
  DDPMIR_HIP_CHECK(hipMemcpyAsync(p.dft[8], x, n * 4, hipMemcpyDeviceToDevice, stream_));
  phase_fix(p.dft[8], B, H, W, alpha, stream_, 0);
  DDPMIR_HIP_CHECK(hipMemcpyAsync(out, p.dft[8], n * 4, hipMemcpyDeviceToDevice, stream_));
  DDPMIR_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Context::jpeg_roundtrip(const uint8_t* in, uint8_t* out, int B, int H, int W, int quality) {
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  ensure_plan(B, H, W);
  order_after_caller();
  launch_jpeg_roundtrip(in, out, plan_.s_jpeg, B, H, W, quality, stream_);
  DDPMIR_HIP_CHECK(hipStreamSynchronize(stream_));
}

// ------------------------------------------------------------------------------------------------ DDRM loop
// One timestep of webp_inference.py:566-600 for micro-batch lane l, in two halves around the codec round trip.  All
// per-step scalars (i, t, the noise scale and key, the last-step flag) live in plan_.sp[l] on the device, and the back
// half ends by advancing them, so both halves are the same launch sequence for every i: they run either eagerly or as
// captured hipGraphs (cfg.use_graph) -- the same kernels with the same arguments, hence bit-identical results.
void Context::step_front(int l, int nb, const LoopConsts& c, hipStream_t s) {
  Plan& p = plan_;
  const size_t nh = (size_t)nb * 3 * p.H * p.W;
  fb_ = nb;
  launch_fill_t_p(p.s_t + l * nb, nb, p.sp + l, s);
  forward(p.s_xt + l * nh, p.s_t + l * nb, p.s_t + l * nb, p.s_theta + l * nh, s);
  launch_quantize_u8(p.s_theta + l * nh, p.s_u8 + l * nh, (int64_t)nh, s);
  if (c.dev_codec)   // JPEG round trip stays on the device: no PCIe hop, no host synchronisation inside the loop
    launch_jpeg_roundtrip(p.s_u8 + l * nh, p.s_dec + l * nh, p.s_jpeg, nb, p.H, p.W, c.quality, s);
  else
    DDPMIR_HIP_CHECK(hipMemcpyAsync(pin_in_ + l * nh, p.s_u8 + l * nh, nh, hipMemcpyDeviceToHost, s));
}

void Context::step_back(int l, int nb, bool with_phase, const LoopConsts& c, hipStream_t s) {
  Plan& p = plan_;
  const size_t nh = (size_t)nb * 3 * p.H * p.W;
  if (!c.dev_codec) DDPMIR_HIP_CHECK(hipMemcpyAsync(p.s_dec + l * nh, pin_out_ + l * nh, nh, hipMemcpyHostToDevice, s));
  // counter = (step, global element index): results do not depend on how the batch is sharded
  launch_randn_p(p.s_noise + l * nh, (int64_t)nh, p.sp + l, s);
  launch_ddrm_update_p(p.s_theta + l * nh, p.s_dec + l * nh, p.s_y + l * nh, p.s_noise + l * nh, (int64_t)(l * nh),
                       p.s_xt + l * nh, (int64_t)nh, c.eta, c.eta_b, c.one_minus_eta_b, p.sp + l, s);
  if (with_phase) phase_fix(p.s_xt + l * nh, nb, p.H, p.W, c.alpha, s, l * nh);
  launch_step_params_advance(p.sp + l, s);
}

template <typename F>
void Context::run_or_replay(GraphSlot& g, bool use_graph, hipStream_t s, F&& body) {
  if (!use_graph || !g.warm) {   // first execution is eager: lazily initialised function attributes must not land in a capture
    body();
    g.warm = true;
    return;
  }
  if (!g.exec) {
    DDPMIR_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try {
      body();
    } catch (...) {
      hipGraph_t broken = nullptr;
      (void)hipStreamEndCapture(s, &broken);
      if (broken) (void)hipGraphDestroy(broken);
      throw;
    }
    DDPMIR_HIP_CHECK(hipStreamEndCapture(s, &g.graph));
    DDPMIR_HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
  }
  DDPMIR_HIP_CHECK(hipGraphLaunch(g.exec, s));
}

void Context::ddrm_sample_loop(const float* y, float* out, int B, int H, int W, int quality, int steps, double eta,
                               double eta_b, const ddpmir_loop_opts* opts) {
  if (!finalized_) throw std::logic_error("sample before finalize_weights");
  if (steps <= 0) throw std::invalid_argument("steps must be positive");
  if (!opts) throw std::invalid_argument("loop options are required");
  LoopConsts c;
  ddpmir_codec_fn codec = opts->codec;
  void* codec_user = opts->codec_user;
  c.dev_codec = codec == nullptr && cfg.arch == DDPMIR_ARCH_JPEG;
  const bool builtin_pool = !codec && !c.dev_codec;
  if (!codec && !c.dev_codec) {   // WebP / AVIF with no callback: the built-in host codec pool (libwebp / libavif C API)
    codec = builtin_host_codec(cfg.arch);
    codec_user = nullptr;
    if (!codec) throw std::runtime_error("codec: no callback given and the built-in host codec library could not be loaded (" +
                                         builtin_host_codec_error() + ")");
  }
  DDPMIR_HIP_CHECK(hipSetDevice(cfg.device));
  if (use_two_lanes(B, H, W, opts, c.dev_codec)) { loop_two_lanes(y, out, B, H, W, quality, steps, eta, eta_b, opts); return; }
  ensure_plan(B, H, W);
  order_after_caller();   // y (and an injected noise tensor) come from the caller's stream
  Plan& p = plan_;
  hipStream_t s = stream_;
  const size_t n = (size_t)B * 3 * H * W;
  c.quality = quality;
  c.eta = (float)eta; c.eta_b = (float)eta_b;
  c.one_minus_eta_b = (float)(1.0 - eta_b);   // Python forms (1 - eta_b) in double (webp_inference.py:592)
  c.alpha = opts->phase_alpha >= 0.0 ? opts->phase_alpha : k.phase_alpha;
  const float coeff = opts->noise_coeff >= 0.0 ? (float)opts->noise_coeff : k.noise_coeff;
  const int phase_q = opts->phase_q >= 0 ? opts->phase_q : k.phase_q;
  const int every = opts->phase_every > 0 ? opts->phase_every : k.phase_every;
  const bool do_phase = quality < phase_q;
  DDPMIR_HIP_CHECK(hipMemcpyAsync(p.s_y, y, n * 4, hipMemcpyDeviceToDevice, s));     // y = x_t.clone()
  DDPMIR_HIP_CHECK(hipMemcpyAsync(p.s_xt, y, n * 4, hipMemcpyDeviceToDevice, s));
  debug_stage(s, "loop", "copy y", B, H, W, 3);
  const int i_first = (opts->start_step >= 0 && opts->start_step < steps) ? opts->start_step : steps - 1;
  const int i_last = opts->num_steps > 0 ? std::max(0, i_first - opts->num_steps + 1) : 0;

  // Host-codec families (WebP / AVIF, or JPEG through the callback): the batch is split into two micro-batch
  // lanes that are software-pipelined -- while the host runs the codec of lane A, the GPU runs the U-Net
  // forward of lane B, and lane A's next forward is enqueued before lane B's codec starts.  Images are
  // independent, so the results are those of the unsplit batch (bit for bit in fp32 mode; in bf16 / fp8 mode up to the
  // mode's noise floor, because kernel selection follows the launch batch).  The device JPEG codec needs no lanes.
  const int nl = (!c.dev_codec && B >= 2 && (B % 2) == 0 && !exp_switch("DDPMIR_NO_OVERLAP")) ? 2 : 1;
  const int nb = B / nl;
  const size_t nh = (size_t)nb * 3 * H * W;                 // elements per lane
  if (do_phase)
    for (int l = 0; l < nl; ++l) phase_prepare(p.s_y + l * nh, nb, H, W, s, l * nh);
  debug_stage(s, "loop", "reference phase of y (fp64 DFT)", B, H, W, 3);
  for (int l = 0; l < nl; ++l)
    launch_step_params_set(p.sp + l, i_first, steps, coeff, opts->seed,
                           (opts->image_offset + (uint64_t)l * nb) * (uint64_t)(3 * H * W) / 4, opts->noise, (int64_t)n, s);
  debug_stage(s, "loop", "step parameters", B, H, W, 3);

  // captured timesteps are valid for one set of baked-in scalars; anything else (seed, noise, offsets, i) is in plan_.sp
  const bool graphs = cfg.use_graph && !prof_on_;   // HIP-event spans cannot bracket nodes of a replayed graph
  if (graphs && !(p.lg_key == c && p.lg_nl == nl)) { drop_loop_graphs(); p.lg_key = c; p.lg_nl = nl; }

  auto front = [&](int l) {   // G(l, i): t, U-Net forward, uint8 quantise, device codec or D2H
    run_or_replay(p.g_front[l], graphs, s, [&] { step_front(l, nb, c, s); });
    if (!c.dev_codec) DDPMIR_HIP_CHECK(hipEventRecord(lane_ev_[l], s));
  };
  std::vector<uint8_t> failed((size_t)nb, 0);
  auto codec_and_update = [&](int l, int i) {  // C(l, i) on the host, then U(l, i) on the stream
    bool any_failed = false;
    if (!c.dev_codec) {
      DDPMIR_HIP_CHECK(hipEventSynchronize(lane_ev_[l]));
      const auto t0 = std::chrono::steady_clock::now();
      int rc;
      if (builtin_pool) {   // per-image status: an image the codec cannot round-trip falls back to JPEG (below), as in the reference
        rc = builtin_host_codec_status(cfg.arch, pin_in_ + l * nh, pin_out_ + l * nh, nb, H, W, quality, failed.data());
        for (int j = 0; j < nb; ++j) any_failed = any_failed || failed[j];
      } else {
        rc = codec(codec_user, pin_in_ + l * nh, pin_out_ + l * nh, nb, H, W, quality);
      }
      codec_host_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      codec_calls_ += 1;
      if (rc != 0) throw std::runtime_error("codec callback failed with code " + std::to_string(rc));
    }
    const bool with_phase = i > 0 && do_phase && (i % every) == 0;
    if (any_failed && cfg.arch != DDPMIR_ARCH_AVIF)   // webp_compress has no fallback: the reference lets the exception propagate
      throw std::runtime_error("codec: the built-in WebP codec failed on an image of the batch");
    if (any_failed) {
      // avif_inference.py:88-95: an image whose AVIF round trip raised is silently round-tripped as JPEG instead (Pillow's default
      // 4:2:0, same quality).  The uint8 image is still on the device: the bit-exact JPEG simulator fills in its decoded pixels
      // after the other images arrive from the host.  (Eager launches for this step: a captured timestep has no such branch.)
      if ((H % 16) || (W % 16)) throw std::runtime_error("codec: per-image JPEG fallback needs image sides that are multiples of 16");
      const size_t img = (size_t)3 * H * W;
      DDPMIR_HIP_CHECK(hipMemcpyAsync(p.s_dec + l * nh, pin_out_ + l * nh, nh, hipMemcpyHostToDevice, s));
      for (int j = 0; j < nb; ++j)
        if (failed[j]) launch_jpeg_roundtrip(p.s_u8 + l * nh + j * img, p.s_dec + l * nh + j * img, p.s_jpeg, 1, H, W, quality, s, 1);
      LoopConsts c2 = c;
      c2.dev_codec = true;   // (step_back: the decoded images are already in s_dec)
      step_back(l, nb, with_phase, c2, s);
    } else
    run_or_replay(p.g_back[l][with_phase ? 1 : 0], graphs, s, [&] { step_back(l, nb, with_phase, c, s); });
    if (opts->trace) {
      float* tr = opts->trace + (size_t)(i_first - i) * 2 * n + l * nh;
      DDPMIR_HIP_CHECK(hipMemcpyAsync(tr, p.s_theta + l * nh, nh * 4, hipMemcpyDeviceToDevice, s));
      DDPMIR_HIP_CHECK(hipMemcpyAsync(tr + n, p.s_xt + l * nh, nh * 4, hipMemcpyDeviceToDevice, s));
    }
  };

  // Two lanes: only the full-resolution blocks run per lane (down1 right after the lane's update, up5 + head right before its
  // codec); the levels below run once for both lanes, so their latency-bound ~2.5 ms is paid once per timestep.  Stream order
  // per timestep: head(0) head(1) mid tail(0) tail(1) | codec(0) on the host under tail(1), codec(1) under the next head(0).
  auto head = [&](int l) {
    run_or_replay(p.g_head[l], graphs, s, [&] {
      launch_fill_t_p(p.s_t + l * nb, nb, p.sp + l, s);
      fwd_head(p.s_xt + l * nh, p.s_t + l * nb, p.s_t + l * nb, l * nb, nb, false, s);
    });
  };
  auto mid = [&] { run_or_replay(p.g_mid, graphs, s, [&] { fwd_mid(B, false, s); }); };
  auto tail = [&](int l) {
    run_or_replay(p.g_tail[l], graphs, s, [&] {
      fwd_tail(p.s_theta + l * nh, l * nb, nb, false, s);
      launch_quantize_u8(p.s_theta + l * nh, p.s_u8 + l * nh, (int64_t)nh, s);
      DDPMIR_HIP_CHECK(hipMemcpyAsync(pin_in_ + l * nh, p.s_u8 + l * nh, nh, hipMemcpyDeviceToHost, s));
    });
    DDPMIR_HIP_CHECK(hipEventRecord(lane_ev_[l], s));
  };
  const bool merged = nl == 2 && !exp_switch("DDPMIR_SEPARATE_LANES");

  try {
    if (merged) {
      head(0); head(1); mid(); tail(0); tail(1);
      for (int i = i_first; i >= i_last; --i) {
        const bool more = i - 1 >= i_last;
        codec_and_update(0, i);
        if (more) head(0);
        codec_and_update(1, i);
        if (more) { head(1); mid(); tail(0); tail(1); }
      }
    } else {
      for (int l = 0; l < nl; ++l) front(l);
      for (int i = i_first; i >= i_last; --i)
        for (int l = 0; l < nl; ++l) {
          codec_and_update(l, i);
          if (i - 1 >= i_last) front(l);
        }
    }
  } catch (...) {
    fb_ = B;
    (void)hipStreamSynchronize(s);
    throw;
  }
  fb_ = B;
  DDPMIR_HIP_CHECK(hipMemcpyAsync(out, p.s_xt, n * 4, hipMemcpyDeviceToDevice, s));
  DDPMIR_HIP_CHECK(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------------------------------------ two concurrent lanes
// Option "concurrent_lanes" (off by default), device-codec JPEG loop, bf16 / fp8 mode, attention in every block.  79 % of a
// timestep is the two full-resolution attention launches, bound by VALU issue (one v_exp_f32 per score) with the matrix pipe a
// third busy and HBM idle; the rest -- convolutions, streaming passes, ~150 latency-bound launches of the deep levels -- leaves
// part of the chip idle (0.65 ms per forward in launches of fewer than 256 workgroups, profiles/r03_jpeg_forward_timeline.txt).  The experiment: split the batch into two
// lanes that run CONCURRENTLY -- this context and a second one that shares its weight blob, each with its own stream and
// workspace (two workspaces for B / 2 = one for B), both enqueued by one host thread; 1 = free-running, 2 = events make the four
// full-resolution attention launches of a timestep take turns so that each runs beside the other lane's non-attention work only.
// Measured (tools/lanes_ab.py, profiles/r04_lanes_ab.txt): free-running x1.005-1.011 on the headline workload, x1.02 at 128 x 128,
// B = 64; taking turns x0.99 -- a measured "no" for the headline, so the default stays one lane.  Images are independent and the
// noise is keyed by the global image index: each lane produces exactly what the single-lane loop produces for its half.
bool Context::use_two_lanes(int B, int H, int W, const ddpmir_loop_opts* opts, bool dev_codec) const {
  if (is_lane_ || lanes_opt_ == 0 || !dev_codec || !use_gemm16() || cfg.use_graph || opts->noise || opts->trace) return false;
  if (B < 2 || (B % 2) != 0 || cfg.attn_min_level != 0) return false;
  // -1 (default) = off: measured on MI355X (profiles/r04_lanes_ab.txt) the lanes buy 0.5-1 % on the headline workload and 2 % at
  // 128 x 128, B = 64 -- the hardware runs the two streams side by side, but the attention launches already fill every CU
  return lanes_opt_ >= 1;
}

void Context::ensure_lane() {
  if (lane_) return;
  std::unique_ptr<Context> c(new Context(cfg));
  c->is_lane_ = true;
  c->attn_nm_bound_ = attn_nm_bound_; c->fp8_min_ch_ = fp8_min_ch_; c->stream16_opt_ = stream16_opt_; c->tile256_opt_ = tile256_opt_; c->epi_direct_opt_ = epi_direct_opt_; c->head_fused_opt_ = head_fused_opt_; c->conv_tile_opt_ = conv_tile_opt_;
  c->finalize_weights(wblob_, 0);   // adopts the packed blob: same layout, no upload
  if (prof_on_) c->profile_enable(prof_detail_);
  lane_ = std::move(c);
}

void Context::loop_two_lanes(const float* y, float* out, int B, int H, int W, int quality, int steps, double eta, double eta_b,
                             const ddpmir_loop_opts* opts) {
  ensure_lane();
  Context* L[2] = {this, lane_.get()};
  const int nb = B / 2;
  const size_t nh = (size_t)nb * 3 * H * W;
  LoopConsts c;
  c.dev_codec = true;
  c.quality = quality;
  c.eta = (float)eta; c.eta_b = (float)eta_b;
  c.one_minus_eta_b = (float)(1.0 - eta_b);
  c.alpha = opts->phase_alpha >= 0.0 ? opts->phase_alpha : k.phase_alpha;
  const float coeff = opts->noise_coeff >= 0.0 ? (float)opts->noise_coeff : k.noise_coeff;
  const int phase_q = opts->phase_q >= 0 ? opts->phase_q : k.phase_q;
  const int every = opts->phase_every > 0 ? opts->phase_every : k.phase_every;
  const bool do_phase = quality < phase_q;
  const int i_first = (opts->start_step >= 0 && opts->start_step < steps) ? opts->start_step : steps - 1;
  const int i_last = opts->num_steps > 0 ? std::max(0, i_first - opts->num_steps + 1) : 0;
  for (int l = 0; l < 2; ++l) {
    Context* X = L[l];
    X->ensure_plan(nb, H, W);
    X->order_after_caller();
    for (auto& e : X->lane_done_)
      if (!e) DDPMIR_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    Plan& p = X->plan_;
    hipStream_t s = X->stream_;
    DDPMIR_HIP_CHECK(hipMemcpyAsync(p.s_y, y + l * nh, nh * 4, hipMemcpyDeviceToDevice, s));
    DDPMIR_HIP_CHECK(hipMemcpyAsync(p.s_xt, y + l * nh, nh * 4, hipMemcpyDeviceToDevice, s));
    if (do_phase) X->phase_prepare(p.s_y, nb, H, W, s, 0);
    launch_step_params_set(p.sp, i_first, steps, coeff, opts->seed, (opts->image_offset + (uint64_t)l * nb) * (uint64_t)(3 * H * W) / 4,
                           nullptr, (int64_t)nh, s);
  }
  auto finish = [&] {
    for (Context* X : L) { X->attn_wait_ = nullptr; X->attn_done_ = nullptr; X->fb_ = X->plan_.B; }
  };
  try {
    for (int i = i_first; i >= i_last; --i) {
      const bool with_phase = i > 0 && do_phase && (i % every) == 0;
      for (int l = 0; l < 2; ++l) {     // down1 of both lanes: lane 0's attention after lane 1's up5 attention of the step before
        Context* X = L[l];
        Plan& p = X->plan_;
        X->attn_wait_ = l == 0 ? (i == i_first ? nullptr : L[1]->lane_done_[1]) : L[0]->lane_done_[0];
        if (lanes_opt_ != 2) X->attn_wait_ = nullptr;   // 1: free-running lanes (the better of the two); 2: attention launches take turns
        X->attn_done_ = X->lane_done_[0];
        launch_fill_t_p(p.s_t, nb, p.sp, X->stream_);
        X->fwd_head(p.s_xt, p.s_t, p.s_t, 0, nb, false, X->stream_);
      }
      for (int l = 0; l < 2; ++l) {     // the levels below, up5 + head, codec, update
        Context* X = L[l];
        Plan& p = X->plan_;
        hipStream_t s = X->stream_;
        X->attn_wait_ = l == 0 ? L[1]->lane_done_[0] : L[0]->lane_done_[1];
        if (lanes_opt_ != 2) X->attn_wait_ = nullptr;
        X->attn_done_ = X->lane_done_[1];
        X->fwd_mid(nb, false, s);
        X->fwd_tail(p.s_theta, 0, nb, false, s);
        launch_quantize_u8(p.s_theta, p.s_u8, (int64_t)nh, s);
        launch_jpeg_roundtrip(p.s_u8, p.s_dec, p.s_jpeg, nb, H, W, quality, s);
        X->step_back(0, nb, with_phase, c, s);
      }
    }
  } catch (...) {
    finish();
    for (Context* X : L) (void)hipStreamSynchronize(X->stream_);
    throw;
  }
  finish();
  for (int l = 0; l < 2; ++l) {
    DDPMIR_HIP_CHECK(hipMemcpyAsync(out + l * nh, L[l]->plan_.s_xt, nh * 4, hipMemcpyDeviceToDevice, L[l]->stream_));
    DDPMIR_HIP_CHECK(hipStreamSynchronize(L[l]->stream_));
  }
}

void Context::drop_loop_graphs() {
  Plan& p = plan_;
  auto drop = [](GraphSlot& g) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g = GraphSlot();
  };
  for (int l = 0; l < 2; ++l) { drop(p.g_front[l]); drop(p.g_back[l][0]); drop(p.g_back[l][1]); drop(p.g_head[l]); drop(p.g_tail[l]); }
  drop(p.g_mid);
}

}  // namespace ddpmir

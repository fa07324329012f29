// extern "C" surface of libddpmir.so (include/ddpmir.h).  Exceptions never cross the boundary.
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "eps_unet.h"
#include "model.h"

using ddpmir::Context;

struct ddpmir_ctx {
  Context* impl = nullptr;          // restoration U-Net context, or
  ddpmir::EpsUnet* eps = nullptr;   // eps-predict UNetDenoiser context
  std::string err;
};

namespace {
thread_local std::string g_create_err;

template <typename F>
int guarded(ddpmir_ctx* c, F&& f, bool want_eps = false) {
  if (!c || (want_eps ? !c->eps : !c->impl)) {
    if (c) c->err = want_eps ? "not an eps-UNet context" : "not a restoration-model context";
    return DDPMIR_EINVAL;
  }
  try {
    f();
    return DDPMIR_OK;
  } catch (const std::invalid_argument& e) {
    c->err = e.what();
    return DDPMIR_EINVAL;
  } catch (const std::logic_error& e) {
    c->err = e.what();
    return DDPMIR_ESTATE;
  } catch (const std::bad_alloc&) {
    c->err = "out of host memory";
    return DDPMIR_ENOMEM;
  } catch (const std::exception& e) {
    c->err = e.what();
    const std::string m = e.what();
    if (m.find("codec") != std::string::npos) return DDPMIR_ECODEC;
    if (m.find("hip") != std::string::npos || m.find("HIP") != std::string::npos) return DDPMIR_EHIP;
    return DDPMIR_EINVAL;
  } catch (...) {
    c->err = "unknown error";
    return DDPMIR_EINVAL;
  }
}
}  // namespace

extern "C" {

ddpmir_ctx* ddpmir_create(const ddpmir_config* cfg) {
  if (!cfg) { g_create_err = "null config"; return nullptr; }
  try {
    auto* c = new ddpmir_ctx();
    c->impl = new Context(*cfg);
    return c;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return nullptr;
  }
}

void ddpmir_destroy(ddpmir_ctx* ctx) {
  if (!ctx) return;
  delete ctx->eps;
  delete ctx->impl;
  delete ctx;
}

const char* ddpmir_last_error(ddpmir_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int ddpmir_load_tensor(ddpmir_ctx* ctx, const char* name, const void* data_host, int dtype, const int64_t* shape, int ndim) {
  const bool e = ctx && ctx->eps;
  return guarded(ctx, [&] {
    if (!name || !data_host || !shape || ndim < 0 || ndim > 8) throw std::invalid_argument("load_tensor: bad argument");
    const float* f32 = static_cast<const float*>(data_host);
    std::vector<float> wide;
    if (dtype != DDPMIR_F32) {   // half-precision checkpoints: widen exactly
      if (dtype != DDPMIR_BF16 && dtype != DDPMIR_F16) throw std::invalid_argument("load_tensor: dtype must be DDPMIR_F32, DDPMIR_BF16 or DDPMIR_F16");
      int64_t n = 1;
      for (int i = 0; i < ndim; ++i) n *= shape[i];
      wide.resize((size_t)std::max<int64_t>(n, 0));
      const uint16_t* h = static_cast<const uint16_t*>(data_host);
      for (int64_t i = 0; i < n; ++i) {
        uint32_t u;
        if (dtype == DDPMIR_BF16) {
          u = (uint32_t)h[i] << 16;
        } else {
          const uint32_t sgn = (uint32_t)(h[i] & 0x8000u) << 16, ex = (h[i] >> 10) & 0x1Fu, man = h[i] & 0x3FFu;
          if (ex == 0) {
            if (man == 0) u = sgn;
            else {   // subnormal half: normalise
              int sh = 0; uint32_t m = man;
              while (!(m & 0x400u)) { m <<= 1; ++sh; }
              u = sgn | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 0x3FFu) << 13);
            }
          } else if (ex == 31) u = sgn | 0x7F800000u | (man << 13);
          else u = sgn | ((ex + 112u) << 23) | (man << 13);
        }
        std::memcpy(&wide[(size_t)i], &u, 4);
      }
      f32 = wide.data();
    }
    if (e) ctx->eps->load_tensor(name, f32, shape, ndim);
    else ctx->impl->load_tensor(name, f32, shape, ndim);
  }, e);
}

int ddpmir_weight_layout_version(void) { return DDPMIR_WEIGHT_LAYOUT_VERSION; }

int ddpmir_set_codec_library(int arch, const char* path) {
  try {
    if (!path) throw std::invalid_argument("set_codec_library: null path");
    ddpmir::set_codec_library(arch, path);
    return DDPMIR_OK;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return DDPMIR_EINVAL;
  }
}

int ddpmir_host_codec_roundtrip_u8(int arch, const uint8_t* in, uint8_t* out, int B, int H, int W, int quality) {
  try {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0) throw std::invalid_argument("host_codec_roundtrip_u8: bad argument");
    ddpmir_codec_fn fn = ddpmir::builtin_host_codec(arch);
    if (!fn) { g_create_err = "codec: " + ddpmir::builtin_host_codec_error(); return DDPMIR_ECODEC; }
    const int rc = fn(nullptr, in, out, B, H, W, quality);
    if (rc != 0) { g_create_err = "codec: the built-in host codec failed with code " + std::to_string(rc); return DDPMIR_ECODEC; }
    return DDPMIR_OK;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return DDPMIR_EINVAL;
  }
}

int ddpmir_host_codec_threads(void) { return ddpmir::builtin_host_codec_threads(); }

size_t ddpmir_weight_blob_bytes(ddpmir_ctx* ctx) {
  size_t n = 0;
  guarded(ctx, [&] { n = ctx->impl->weight_blob_bytes(); });
  return n;
}

int ddpmir_finalize_weights(ddpmir_ctx* ctx, void* dev_blob, int fill) {
  const bool e = ctx && ctx->eps;
  return guarded(ctx, [&] {
    if (e) ctx->eps->finalize();
    else ctx->impl->finalize_weights(dev_blob, fill);
  }, e);
}

int64_t ddpmir_param_count(ddpmir_ctx* ctx) { return (ctx && ctx->impl) ? ctx->impl->param_count() : -1; }

int ddpmir_predict(ddpmir_ctx* ctx, const float* x, const float* t, const float* cl, float* out, int B, int H, int W,
                   void* stream) {
  return guarded(ctx, [&] {
    if (!x || !t || !out) throw std::invalid_argument("predict: null pointer");
    ctx->impl->predict(x, t, cl, out, B, H, W, static_cast<hipStream_t>(stream));
  });
}

int ddpmir_ddrm_sample_loop(ddpmir_ctx* ctx, const float* y, float* out, int B, int H, int W, int quality, int steps,
                            double eta, double eta_b, const ddpmir_loop_opts* opts) {
  return guarded(ctx, [&] {
    if (!y || !out) throw std::invalid_argument("sample: null pointer");
    ctx->impl->ddrm_sample_loop(y, out, B, H, W, quality, steps, eta, eta_b, opts);
  });
}

int ddpmir_phase_consistency(ddpmir_ctx* ctx, const float* x, const float* ref, float* out, int B, int H, int W,
                             double alpha) {
  return guarded(ctx, [&] {
    if (!x || !ref || !out) throw std::invalid_argument("phase_consistency: null pointer");
    ctx->impl->phase_consistency(x, ref, out, B, H, W, alpha);
  });
}

int ddpmir_profile_enable(ddpmir_ctx* ctx, int enable) {
  return guarded(ctx, [&] { ctx->impl->profile_enable(enable); });
}

int ddpmir_profile_read(ddpmir_ctx* ctx, const char* cls, double* ms, int64_t* launches, double* flops) {
  return guarded(ctx, [&] {
    if (!cls) throw std::invalid_argument("profile_read: null class");
    ctx->impl->profile_read(cls, ms, launches, flops);
  });
}

#ifndef DDPMIR_SRC_HASH
#define DDPMIR_SRC_HASH "unknown"
#endif
#ifndef DDPMIR_HIPCC_VERSION
#define DDPMIR_HIPCC_VERSION "unknown"
#endif
const char* ddpmir_version(void) { return "ddpmir 0.4.0 (gfx950) src=" DDPMIR_SRC_HASH " hipcc=" DDPMIR_HIPCC_VERSION; }

int ddpmir_set_option(ddpmir_ctx* ctx, const char* name, double value) {
  return guarded(ctx, [&] {
    if (!name) throw std::invalid_argument("set_option: null name");
    ctx->impl->set_option(name, value);
  });
}

}  // extern "C"

extern "C" {

ddpmir_ctx* ddpmir_create_eps_unet(int device) {
  try {
    auto* c = new ddpmir_ctx();
    c->eps = new ddpmir::EpsUnet(device);
    return c;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return nullptr;
  }
}

int ddpmir_predict_noise(ddpmir_ctx* ctx, const float* x, float t, float* eps, int B, int H, int W) {
  return guarded(ctx, [&] {
    if (!x || !eps) throw std::invalid_argument("predict_noise: null pointer");
    ctx->eps->predict_sync(x, t, eps, B, H, W);
  }, true);
}

int ddpmir_p_sample_loop(ddpmir_ctx* ctx, const float* x, float* out, int B, int H, int W, int T,
                         const float* betas_host, int n_betas) {
  return guarded(ctx, [&] {
    if (!x || !out || !betas_host) throw std::invalid_argument("p_sample_loop: null pointer");
    ctx->eps->p_sample_loop(x, out, B, H, W, T, betas_host, n_betas);
  }, true);
}

}  // extern "C"

extern "C" int ddpmir_jpeg_roundtrip_u8(ddpmir_ctx* ctx, const uint8_t* in, uint8_t* out, int B, int H, int W, int quality) {
  return guarded(ctx, [&] {
    if (!in || !out) throw std::invalid_argument("jpeg_roundtrip_u8: null pointer");
    ctx->impl->jpeg_roundtrip(in, out, B, H, W, quality);
  });
}

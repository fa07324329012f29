// extern "C" surface of libddpmir.so (include/ddpmir.h).  Exceptions never cross the boundary.
#include <cstring>
#include <stdexcept>
#include <string>

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

int ddpmir_load_tensor(ddpmir_ctx* ctx, const char* name, const float* data_host, const int64_t* shape, int ndim) {
  const bool e = ctx && ctx->eps;
  return guarded(ctx, [&] {
    if (!name || !data_host || !shape || ndim < 0 || ndim > 8) throw std::invalid_argument("load_tensor: bad argument");
    if (e) ctx->eps->load_tensor(name, data_host, shape, ndim);
    else ctx->impl->load_tensor(name, data_host, shape, ndim);
  }, e);
}

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

const char* ddpmir_version(void) { return "ddpmir 0.1.0 (gfx950)"; }

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

// eps-predict DDPM flavour: the early UNetDenoiser (experiments/code/conv.ipynb c0:42-101) and the deterministic
// posterior-mean loop ddpm_inference (conv.ipynb c1:113-125) -- the only place in the reference with literal
// predict_noise / p_sample semantics.  fp32, built from the same implicit-GEMM and resampling kernels.
#include <cmath>
#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

#include "eps_unet.h"

namespace ddpmir {

namespace {

__global__ void eps_time_embed_k(float t, const float* __restrict__ w0, const float* __restrict__ b0,
                                 const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ out) {
  __shared__ float h[512];
  const int tid = threadIdx.x;   // 512 threads
  h[tid] = fmaxf(fmaf(w0[tid], t, b0[tid]), 0.f);          // Linear(1,512) + ReLU
  __syncthreads();
  const float* wr = w2 + (size_t)tid * 512;
  float acc = 0.f;
  for (int k = 0; k < 512; ++k) acc = fmaf(wr[k], h[k], acc);
  out[tid] = acc + b2[tid];
}

// x[m][c] += e[c]
__global__ void add_channel_vec_k(float* x, const float* e, int64_t M, int C) {
  const int64_t total = M * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    x[i] += e[i % C];
}

// ConvTranspose2d(k=2,s=2) second half: tmp[m][(dy*2+dx)*C + c] (+bias) -> out[b][2y+dy][2x+dx][c] (stride ldo)
__global__ void pixel_shuffle2_k(const float* __restrict__ tmp, const float* __restrict__ bias, float* __restrict__ out,
                                 int ldo, int B, int H, int W, int C) {
  const int64_t total = (int64_t)B * H * W * 4 * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = i % C;
    int64_t t = i / C;
    int tap = t & 3;
    int64_t m = t >> 2;
    int x = m % W;
    int64_t r = m / W;
    int y = r % H;
    int b = r / H;
    int dy = tap >> 1, dx = tap & 1;
    out[(((size_t)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * ldo + c] = tmp[i] + bias[c];
  }
}

inline unsigned gridf(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

EpsUnet::EpsUnet(int device) : device_(device) {
  DDPMIR_HIP_CHECK(hipSetDevice(device));
  DDPMIR_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
}

EpsUnet::~EpsUnet() {
  (void)hipSetDevice(device_);
  for (void* p : allocs_) (void)hipFree(p);
  if (ws_) (void)hipFree(ws_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

void EpsUnet::load_tensor(const std::string& name, const float* data, const int64_t* shape, int ndim) {
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  host_[name] = std::move(t);
}

float* EpsUnet::upload(const std::vector<float>& v) {
  void* d = nullptr;
  DDPMIR_HIP_CHECK(hipMalloc(&d, v.size() * 4 + 16));
  DDPMIR_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  allocs_.push_back(d);
  return static_cast<float*>(d);
}

const HostTensor& EpsUnet::need(const std::string& n, int64_t numel) {
  auto it = host_.find(n);
  if (it == host_.end()) throw std::runtime_error("missing tensor in state dict: " + n);
  if (it->second.numel() != numel) throw std::runtime_error("shape mismatch for " + n);
  return it->second;
}

EpsUnet::Conv EpsUnet::conv3(const std::string& name, int cin, int cout, int k) {
  const auto& w = need(name + ".weight", (int64_t)cout * cin * k * k);
  const auto& b = need(name + ".bias", cout);
  Conv c;
  c.cin = cin; c.cinp = round_up(cin, 4); c.cout = cout; c.coutp = round_up(cout, 4); c.taps = k * k;
  std::vector<float> packed((size_t)c.taps * c.coutp * c.cinp, 0.f), bias(c.coutp, 0.f);
  for (int o = 0; o < cout; ++o) {
    bias[o] = b.data[o];
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < c.taps; ++t)
        packed[((size_t)t * c.coutp + o) * c.cinp + ci] = w.data[((size_t)o * cin + ci) * c.taps + t];
  }
  c.w = upload(packed); c.b = upload(bias);
  return c;
}

// ConvTranspose2d weight [Cin][Cout][2][2] -> GEMM weight [(dy*2+dx)*Cout + co][ci]
EpsUnet::Conv EpsUnet::convT(const std::string& name, int cin, int cout) {
  const auto& w = need(name + ".weight", (int64_t)cin * cout * 4);
  const auto& b = need(name + ".bias", cout);
  Conv c;
  c.cin = cin; c.cinp = cin; c.cout = 4 * cout; c.coutp = 4 * cout; c.taps = 1;
  std::vector<float> packed((size_t)4 * cout * cin);
  for (int ci = 0; ci < cin; ++ci)
    for (int co = 0; co < cout; ++co)
      for (int t = 0; t < 4; ++t) packed[((size_t)t * cout + co) * cin + ci] = w.data[((size_t)ci * cout + co) * 4 + t];
  c.w = upload(packed);
  c.b = upload(b.data);
  return c;
}

void EpsUnet::finalize() {
  DDPMIR_HIP_CHECK(hipSetDevice(device_));
  const char* dc[7] = {"encoder1", "encoder2", "encoder3", "bottleneck", "decoder3", "decoder2", "decoder1"};
  const int cin[7] = {3, 64, 128, 256, 512, 256, 128}, cout[7] = {64, 128, 256, 512, 256, 128, 64};
  for (int i = 0; i < 7; ++i) {
    dc_[i][0] = conv3(std::string(dc[i]) + ".conv.0", cin[i], cout[i], 3);
    dc_[i][1] = conv3(std::string(dc[i]) + ".conv.2", cout[i], cout[i], 3);
  }
  up_[0] = convT("up3", 512, 256); up_[1] = convT("up2", 256, 128); up_[2] = convT("up1", 128, 64);
  out_conv_ = conv3("out_conv", 64, 3, 1);
  te_w0_ = upload(need("time_embed.0.weight", 512).data); te_b0_ = upload(need("time_embed.0.bias", 512).data);
  te_w2_ = upload(need("time_embed.2.weight", 512 * 512).data); te_b2_ = upload(need("time_embed.2.bias", 512).data);
  host_.clear();
  ready_ = true;
}

void EpsUnet::run_conv(const Conv& c, const float* in, int ldin, int B, int H, int W, float* out, int ldo, int act,
                       bool bias_in_gemm) {
  IGemm p;
  p.A = in; p.lda = ldin; p.B = B; p.H = H; p.W = W; p.Cin = c.cinp; p.Wt = c.w; p.Cout = c.coutp; p.taps = c.taps;
  p.bias = bias_in_gemm ? c.b : nullptr;     // transposed convs add their bias after the pixel shuffle
  p.epi_act = act; p.out = out; p.ldo = ldo;
  launch_igemm(p, stream_);
}

void EpsUnet::predict(const float* x, float t, float* eps, int B, int H, int W) {
  if (!ready_) throw std::logic_error("predict_noise before finalize_weights");
  if (H % 4 || W % 4) throw std::invalid_argument("eps-UNet: H and W must be multiples of 4");
  DDPMIR_HIP_CHECK(hipSetDevice(device_));
  // workspace
  const size_t M0 = (size_t)B * H * W;
  size_t need_f = M0 * 1800 + 16384;      // sum of the buffers taken below (1792 floats per input pixel)
  if (need_f > ws_floats_) {
    if (ws_) (void)hipFree(ws_);
    DDPMIR_HIP_CHECK(hipMalloc(&ws_, need_f * 4));
    ws_floats_ = need_f;
  }
  hipStream_t s = stream_;
  float* w = ws_;
  auto take = [&](size_t n) { float* p = w; w += (n + 63) & ~size_t(63); return p; };
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
  float* x4 = take(M0 * 4);
  float* tmpA = take(M0 * 4 * 64);                    // largest intermediate: decoder1 runs at 2H x 2W
  float* x1 = take(M0 * 64);
  float* p1 = take(M0 / 4 * 64);
  float* x2 = take(M0 / 4 * 128);
  float* p2 = take(M0 / 16 * 128);
  float* x3 = take(M0 / 16 * 256);
  float* bt = take(M0 / 16 * 512);
  float* temb = take(512);
  float* cat3 = take(M0 / 4 * 512);                   // at H2: [x3 resized 256 | up3 256]
  float* d3 = take(M0 / 4 * 256);
  float* cat2 = take(M0 * 256);                       // at H: [x2 resized 128 | up2 128]
  float* d2 = take(M0 * 128);
  float* cat1 = take(M0 * 4 * 128);                   // at 2H: [x1 resized 64 | up1 64]
  float* d1 = take(M0 * 4 * 64);
  float* o4 = take(M0 * 4 * 4);
  float* o4s = take(M0 * 4);
  (void)need_f;

  launch_nchw3_to_nhwc4(x, x4, B, H, W, s);
  // encoders (conv3x3 + ReLU twice)                                   conv.ipynb c0:82-84
  run_conv(dc_[0][0], x4, 4, B, H, W, tmpA, 64, ACT_RELU);
  run_conv(dc_[0][1], tmpA, 64, B, H, W, x1, 64, ACT_RELU);
  launch_maxpool2(x1, 64, p1, 64, B, H, W, 64, s);
  run_conv(dc_[1][0], p1, 64, B, H2, W2, tmpA, 128, ACT_RELU);
  run_conv(dc_[1][1], tmpA, 128, B, H2, W2, x2, 128, ACT_RELU);
  launch_maxpool2(x2, 128, p2, 128, B, H2, W2, 128, s);
  run_conv(dc_[2][0], p2, 128, B, H4, W4, tmpA, 256, ACT_RELU);
  run_conv(dc_[2][1], tmpA, 256, B, H4, W4, x3, 256, ACT_RELU);
  // bottleneck + time embedding                                        conv.ipynb c0:79-80,86
  run_conv(dc_[3][0], x3, 256, B, H4, W4, tmpA, 512, ACT_RELU);
  run_conv(dc_[3][1], tmpA, 512, B, H4, W4, bt, 512, ACT_RELU);
  hipLaunchKernelGGL(eps_time_embed_k, dim3(1), dim3(512), 0, s, t, te_w0_, te_b0_, te_w2_, te_b2_, temb);
  hipLaunchKernelGGL(add_channel_vec_k, dim3(gridf((int64_t)B * H4 * W4 * 512)), dim3(256), 0, s, bt, temb,
                     (int64_t)B * H4 * W4, 512);
  // up3: ConvTranspose 512->256 to H2; cat([resize(x3), up])          conv.ipynb c0:88-90
  run_conv(up_[0], bt, 512, B, H4, W4, tmpA, 4 * 256, ACT_NONE, false);
  hipLaunchKernelGGL(pixel_shuffle2_k, dim3(gridf((int64_t)B * H4 * W4 * 4 * 256)), dim3(256), 0, s, tmpA, up_[0].b,
                     cat3 + 256, 512, B, H4, W4, 256);
  launch_bilinear(x3, 256, cat3, 512, B, H4, W4, H2, W2, 256, 0.5f, 0.5f, 0, 1.0f, s);
  run_conv(dc_[4][0], cat3, 512, B, H2, W2, tmpA, 256, ACT_RELU);
  run_conv(dc_[4][1], tmpA, 256, B, H2, W2, d3, 256, ACT_RELU);
  // up2 to H
  run_conv(up_[1], d3, 256, B, H2, W2, tmpA, 4 * 128, ACT_NONE, false);
  hipLaunchKernelGGL(pixel_shuffle2_k, dim3(gridf((int64_t)B * H2 * W2 * 4 * 128)), dim3(256), 0, s, tmpA, up_[1].b,
                     cat2 + 128, 256, B, H2, W2, 128);
  launch_bilinear(x2, 128, cat2, 256, B, H2, W2, H, W, 128, 0.5f, 0.5f, 0, 1.0f, s);
  run_conv(dc_[5][0], cat2, 256, B, H, W, tmpA, 128, ACT_RELU);
  run_conv(dc_[5][1], tmpA, 128, B, H, W, d2, 128, ACT_RELU);
  // up1 to 2H
  run_conv(up_[2], d2, 128, B, H, W, tmpA, 4 * 64, ACT_NONE, false);
  hipLaunchKernelGGL(pixel_shuffle2_k, dim3(gridf((int64_t)B * H * W * 4 * 64)), dim3(256), 0, s, tmpA, up_[2].b,
                     cat1 + 64, 128, B, H, W, 64);
  launch_bilinear(x1, 64, cat1, 128, B, H, W, 2 * H, 2 * W, 64, 0.5f, 0.5f, 0, 1.0f, s);
  run_conv(dc_[6][0], cat1, 128, B, 2 * H, 2 * W, tmpA, 64, ACT_RELU);
  run_conv(dc_[6][1], tmpA, 64, B, 2 * H, 2 * W, d1, 64, ACT_RELU);
  // out_conv 1x1 -> bilinear back to the input size -> tanh           conv.ipynb c0:101 (size=(32,32) there)
  run_conv(out_conv_, d1, 64, B, 2 * H, 2 * W, o4, 4, ACT_NONE);
  launch_bilinear(o4, 4, o4s, 4, B, 2 * H, 2 * W, H, W, 4, 2.0f, 2.0f, 0, 1.0f, s);
  launch_nhwc4_to_nchw3_tanh(o4s, eps, B, H, W, s);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void EpsUnet::predict_sync(const float* x, float t, float* eps, int B, int H, int W) {
  predict(x, t, eps, B, H, W);
  DDPMIR_HIP_CHECK(hipStreamSynchronize(stream_));
}

// ddpm_inference: x <- (x - (1-a_t)/sqrt(1-abar_t) * eps(x,t)) / sqrt(a_t), t = T..1, no noise term
void EpsUnet::p_sample_loop(const float* x, float* out, int B, int H, int W, int T, const float* betas, int n_betas) {
  if (T < 1 || T > n_betas) throw std::invalid_argument("p_sample_loop: need 1 <= T <= len(betas)");
  DDPMIR_HIP_CHECK(hipSetDevice(device_));
  const size_t n = (size_t)B * 3 * H * W;
  float *xt = nullptr, *eps = nullptr;
  DDPMIR_HIP_CHECK(hipMalloc(&xt, n * 4));
  DDPMIR_HIP_CHECK(hipMalloc(&eps, n * 4));
  try {
    DDPMIR_HIP_CHECK(hipMemcpyAsync(xt, x, n * 4, hipMemcpyDeviceToDevice, stream_));
    for (int t = T; t >= 1; --t) {
      predict(xt, (float)t, eps, B, H, W);
      // fp32 like torch: alpha = 1 - beta (fp32); alpha_bar = prod(alpha[:t]) in fp32, sequential
      float alpha = 1.0f - betas[t - 1];
      float abar = 1.0f;
      for (int i = 0; i < t; ++i) abar *= (1.0f - betas[i]);
      float c1 = 1.0f / std::sqrt(alpha);
      float c2 = (1.0f - alpha) / std::sqrt(1.0f - abar);
      launch_axpby(xt, eps, xt, c1, c2, (int64_t)n, stream_);
    }
    DDPMIR_HIP_CHECK(hipMemcpyAsync(out, xt, n * 4, hipMemcpyDeviceToDevice, stream_));
    DDPMIR_HIP_CHECK(hipStreamSynchronize(stream_));
  } catch (...) {
    (void)hipFree(xt); (void)hipFree(eps);
    throw;
  }
  (void)hipFree(xt); (void)hipFree(eps);
}

}  // namespace ddpmir

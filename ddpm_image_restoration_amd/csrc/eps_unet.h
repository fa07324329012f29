// eps-predict UNetDenoiser + DDPM posterior-mean loop (experiments/code/conv.ipynb).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "model.h"

namespace ddpmir {

class EpsUnet {
 public:
  explicit EpsUnet(int device);
  ~EpsUnet();
  void load_tensor(const std::string& name, const float* data, const int64_t* shape, int ndim);
  void finalize();
  void predict_sync(const float* x, float t, float* eps, int B, int H, int W);
  void p_sample_loop(const float* x, float* out, int B, int H, int W, int T, const float* betas, int n_betas);

 private:
  struct Conv { float* w = nullptr; float* b = nullptr; int cin = 0, cinp = 0, cout = 0, coutp = 0, taps = 1; };
  int device_;
  bool ready_ = false;
  hipStream_t stream_ = nullptr;
  std::map<std::string, HostTensor> host_;
  std::vector<void*> allocs_;
  Conv dc_[7][2], up_[3], out_conv_;
  float *te_w0_ = nullptr, *te_b0_ = nullptr, *te_w2_ = nullptr, *te_b2_ = nullptr;
  float* ws_ = nullptr; size_t ws_floats_ = 0;
  float* upload(const std::vector<float>& v);
  const HostTensor& need(const std::string& n, int64_t numel);
  Conv conv3(const std::string& name, int cin, int cout, int k);
  Conv convT(const std::string& name, int cin, int cout);
  void run_conv(const Conv& c, const float* in, int ldin, int B, int H, int W, float* out, int ldo, int act,
                bool bias_in_gemm = true);
  void predict(const float* x, float t, float* eps, int B, int H, int W);
};

}  // namespace ddpmir

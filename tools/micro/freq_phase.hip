// Where does a tile of the fused frequency-gate kernel spend its life?  Includes the product kernel with DDPMIR_FREQ_PROBE
// (s_memtime stamps by thread 0 at the phase boundaries) and prints the mean duration of each phase over the tiles of the middle
// half of the grid, the tile lifetime and the kernel's duration (HIP events) -- 512 x 512, B = 8, C = 64, 8 x 8 blocks, bf16 in / out.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ../../ddpm_image_restoration_amd/csrc -o freq_phase freq_phase.hip && ./freq_phase
#define DDPMIR_FREQ_PROBE 1
#include "../../ddpm_image_restoration_amd/csrc/freq_fused.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = 512, W = 512, C = 64, bs = 8, low = 4;
  const bool in16 = argc > 2 ? atoi(argv[2]) != 0 : true;
  const size_t n = (size_t)B * H * W * C;
  std::vector<unsigned short> hx(n);
  std::vector<float> hx32(in16 ? 0 : n);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    const float v = (rand() % 2001 - 1000) * 1e-3f;
    unsigned u; memcpy(&u, &v, 4); hx[i] = (unsigned short)(u >> 16);
    if (!in16) hx32[i] = v;
  }
  std::vector<unsigned short> hw1((size_t)C * C), hw2((size_t)2 * C * C);
  auto rb = [&]() { const float v = (rand() % 2001 - 1000) * 1e-4f; unsigned u; memcpy(&u, &v, 4); return (unsigned short)(u >> 16); };
  for (auto& v : hw1) v = rb();
  for (auto& v : hw2) v = rb();
  std::vector<float> hd(64), hb(2 * C, 0.01f), hboost(B, 1.5f);
  for (int u = 0; u < 8; ++u) for (int k = 0; k < 8; ++k) hd[u * 8 + k] = (u == 0 ? 0.35355339f : 0.5f * cosf((2 * k + 1) * u * 3.14159265f / 16));
  void *x16 = nullptr, *x32 = nullptr, *w1, *w2, *S; float *dct, *b1, *b2, *boost; unsigned long long* probe;
  if (in16) { CK(hipMalloc(&x16, n * 2)); CK(hipMemcpy(x16, hx.data(), n * 2, hipMemcpyHostToDevice)); }
  else { CK(hipMalloc(&x32, n * 4)); CK(hipMemcpy(x32, hx32.data(), n * 4, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&w1, hw1.size() * 2)); CK(hipMalloc(&w2, hw2.size() * 2)); CK(hipMalloc(&S, n * 2));
  CK(hipMalloc(&dct, 256)); CK(hipMalloc(&b1, C * 4)); CK(hipMalloc(&b2, 2 * C * 4)); CK(hipMalloc(&boost, B * 4));
  CK(hipMemcpy(w1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dct, hd.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b1, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b2, hb.data(), 2 * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(boost, hboost.data(), B * 4, hipMemcpyHostToDevice));
  const size_t tiles = (size_t)B * H * W / 128;
  CK(hipMalloc(&probe, tiles * 32 * 8)); CK(hipMemset(probe, 0, tiles * 32 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ddpmir::g_probe), &probe, sizeof(probe)));
  auto run = [&]() { ddpmir::launch_freq_gate_fused((const float*)x32, dct, w1, b1, w2, b2, boost, S, B, H, W, C, bs, low, nullptr, nullptr, nullptr, nullptr, nullptr, x16); };
  for (int i = 0; i < 3; ++i) run();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) run();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)n * ((in16 ? 2 : 4) + 2);
  printf("B=%d in=%s: %.1f us per launch (with the probes' stores), %.2f TB/s algorithmic\n", B, in16 ? "bf16" : "fp32", ms * 100, bytes / (ms * 1e-4) * 1e-12);
  std::vector<unsigned long long> hp(tiles * 32);
  CK(hipMemcpy(hp.data(), probe, tiles * 32 * 8, hipMemcpyDeviceToHost));
  double sum[9] = {0}; size_t cnt = 0;
  for (size_t t = tiles / 4; t < 3 * tiles / 4; ++t) {
    const unsigned long long* q = &hp[t * 32];
    for (int k = 1; k < 9; ++k) sum[k] += (double)(q[k] - q[k - 1]);
    ++cnt;
  }
  const char* nm[9] = {"", "A load x -> LDS", "B1 DCT pass 1", "B2 DCT pass 2 + operand rows", "C GEMM1 (W1 from L2)", "C' hidden rows -> LDS",
                       "D GEMM2 (W2 from L2)", "E half 0 (gate, S, store)", "E half 1"};
  double tot = 0;
  for (int k = 1; k < 9; ++k) tot += sum[k] / cnt;
  printf("phase means over %zu tiles, s_memtime ticks (shader clock; the line below the table gives its rate)\n", cnt);
  for (int k = 1; k < 9; ++k) printf("  %-32s %8.1f  (%4.1f %%)\n", nm[k], sum[k] / cnt, 100 * sum[k] / cnt / tot);
  printf("  %-32s %8.1f\n", "tile lifetime", tot);
  // per wave: how long after the first wave of its workgroup it entered, and entry -> x loads returned
  double skew = 0, ld = 0, ldmax = 0;
  for (size_t t = tiles / 4; t < 3 * tiles / 4; ++t) {
    const unsigned long long* q = &hp[t * 32 + 16];
    unsigned long long e0 = q[0], e1 = q[0];
    double l = 0, lm = 0;
    for (int w = 0; w < 4; ++w) {
      e0 = q[2 * w] < e0 ? q[2 * w] : e0; e1 = q[2 * w] > e1 ? q[2 * w] : e1;
      const double d = (double)(q[2 * w + 1] - q[2 * w]);
      l += d / 4; lm = d > lm ? d : lm;
    }
    skew += (double)(e1 - e0); ld += l; ldmax += lm;
  }
  // resident tiles: sum of lifetimes / (last end - first start), on the constant 100 MHz counter (s_memrealtime)
  {
    unsigned long long lo = ~0ull, hi = 0; double life = 0, life_t = 0;
    for (size_t t = 0; t < tiles; ++t) { const unsigned long long* q = &hp[t * 32]; lo = q[24] < lo ? q[24] : lo; hi = q[25] > hi ? q[25] : hi; life += (double)(q[25] - q[24]); life_t += (double)(q[8] - q[0]); }
    printf("  first start -> last end %.1f us; tile lifetime %.2f us = %.0f ticks -> %.2f GHz; %.0f tiles resident on average (256 CUs x 3 = 768 slots)\n",
           (hi - lo) * 0.01, life / tiles * 0.01, life_t / tiles, life_t / life * 0.1, life / (double)(hi - lo));
  }
  printf("  entry skew between the 4 waves of a tile %.1f; entry -> x loads returned: mean %.1f, slowest wave %.1f\n", skew / cnt, ld / cnt, ldmax / cnt);
  return 0;
}

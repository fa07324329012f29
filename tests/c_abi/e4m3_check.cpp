// Host-side check of csrc/common.h f32_to_e4m3 (the weight packer of fp8 mode; host twin of v_cvt_pk_fp8_f32): for every test
// value the result must be the OCP e4m3fn code nearest to it (ties to the even code, saturating at +-448, NaN -> 0x7F), found here
// by brute force over the 127 non-negative finite codes.  Built and run by tests/test_host_logic.py (no GPU).
#include "common.h"
#include <cmath>
#include <cstdio>
#include <random>
static float decode(unsigned c) {          // c in 0 .. 0x7E
  const int e = (c >> 3) & 15, m = c & 7;
  return e == 0 ? std::ldexp((float)m, -9) : std::ldexp((float)(8 + m), e - 10);
}
static unsigned char nearest(float f) {
  const unsigned char sign = std::signbit(f) ? 0x80 : 0;
  const double a = std::fabs((double)f);
  if (a != a) return sign | 0x7F;
  unsigned best = 0; double bd = 1e300;
  for (unsigned c = 0; c <= 0x7E; ++c) {
    const double d = std::fabs((double)decode(c) - a);
    if (d < bd || (d == bd && (c & 1) == 0)) { best = c; bd = d; }
  }
  if (a >= 448.0) best = 0x7E;
  return (unsigned char)(sign | best);
}
int main() {
  long long n = 0;
  auto check = [&](float f) {
    ++n;
    const unsigned char got = ddpmir::f32_to_e4m3(f), want = nearest(f);
    if (got != want && !(f == 0.f && (got & 0x7F) == 0 && (want & 0x7F) == 0)) { printf("FAIL %a (%g): got 0x%02X want 0x%02X\n", f, f, got, want); return false; }
    return true;
  };
  for (unsigned c = 0; c <= 0x7E; ++c) {   // every code, its neighbours' midpoints and one ulp either side of them
    const float v = decode(c), w = decode(c < 0x7E ? c + 1 : c), mid = 0.5f * (v + w);
    for (float f : {v, mid, std::nextafter(mid, 0.f), std::nextafter(mid, 1e9f), std::nextafter(v, 0.f), std::nextafter(v, 1e9f)})
      for (float s : {1.f, -1.f}) if (!check(s * f)) return 1;
  }
  std::mt19937 rng(3);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  for (int i = 0; i < 400000; ++i) {
    const float f = u(rng) * std::ldexp(1.f, (int)(rng() % 22) - 12);    // magnitudes 2^-12 .. 2^9
    if (!check(f)) return 1;
  }
  for (float f : {448.f, 449.f, 464.f, 480.f, 1e6f, -1e6f, INFINITY, -INFINITY, 1e-12f, -1e-12f, 0.f, -0.f}) if (!check(f)) return 1;
  if ((ddpmir::f32_to_e4m3(NAN) & 0x7F) != 0x7F) { printf("FAIL NaN\n"); return 1; }
  printf("E4M3-OK %lld values\n", n);
  return 0;
}

// Host-side check of csrc/common.h FastDiv (the magic-number division the GEMM and pointwise kernels use for pixel index ->
// (sample, row, column)): q = div(n) must equal n / d for every n < 2^31.  The device path differs only in __umulhi vs the
// 64-bit product, which are the same function.  Built and run by tests/test_host_logic.py (hipcc, host code only, no GPU).
#include "common.h"
#include <cstdio>
#include <cstdint>
#include <random>
int main() {
  using ddpmir::FastDiv;
  const unsigned ds[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 24, 32, 48, 64, 96, 100, 128, 160, 192, 224, 255, 256, 320, 512, 1000, 1024, 4096,
                         6144, 16384, 61440, 65536, 262144, 1048576, 2097152, 16777216, 1u << 30, (1u << 31) - 1, 1u << 31};
  std::mt19937_64 rng(7);
  long long checked = 0;
  for (unsigned d : ds) {
    const FastDiv f(d);
    auto check = [&](unsigned n) {
      ++checked;
      if (f.div(n) != n / d) { printf("FAIL d=%u n=%u got %u want %u\n", d, n, f.div(n), n / d); return false; }
      return true;
    };
    for (unsigned n = 0; n < 70000; ++n) if (!check(n)) return 1;                       // small n exhaustively
    for (unsigned long long k = 0; k * d < (1ull << 31); k += 1 + ((1ull << 31) / d) / 50000) {   // around multiples of d
      const unsigned long long m = k * d;
      for (long long e = -2; e <= 2; ++e) { const long long n = (long long)m + e; if (n >= 0 && n < (1ll << 31) && !check((unsigned)n)) return 1; }
    }
    for (int i = 0; i < 200000; ++i) if (!check((unsigned)(rng() & 0x7FFFFFFFu))) return 1;
    if (!check(0x7FFFFFFFu)) return 1;
  }
  printf("FASTDIV-OK %lld divisions\n", checked);
  return 0;
}

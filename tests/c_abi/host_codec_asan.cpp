// Driver for the AddressSanitizer / UBSan build of the host codec pool (CPU only, gcc): runs batches through both built-in
// codecs from two threads at once and prints a checksum.   usage: host_codec_asan <libwebp path> <libavif path>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../ddpm_image_restoration_amd/csrc/host_codec.h"

static unsigned run(int arch, int B, int H, int W, int quality) {
  std::vector<uint8_t> in((size_t)B * 3 * H * W), out(in.size());
  for (size_t i = 0; i < in.size(); ++i) in[i] = (uint8_t)((i * 2654435761u) >> 24);
  ddpmir_codec_fn fn = ddpmir::builtin_host_codec(arch);
  if (!fn) { std::printf("arch %d unavailable: %s\n", arch, ddpmir::builtin_host_codec_error().c_str()); return 0; }
  if (fn(nullptr, in.data(), out.data(), B, H, W, quality) != 0) { std::printf("arch %d failed\n", arch); return 0; }
  unsigned sum = 0;
  for (uint8_t v : out) sum = sum * 31u + v;
  return sum;
}

int main(int argc, char** argv) {
  if (argc > 1) ddpmir::set_codec_library(DDPMIR_ARCH_WEBP, argv[1]);
  if (argc > 2) ddpmir::set_codec_library(DDPMIR_ARCH_AVIF, argv[2]);
  unsigned a = 0, b = 0;
  std::thread t1([&] { a = run(DDPMIR_ARCH_WEBP, 5, 40, 72, 10); });
  std::thread t2([&] { b = run(DDPMIR_ARCH_AVIF, 3, 32, 64, 10); });
  t1.join(); t2.join();
  const unsigned a2 = run(DDPMIR_ARCH_WEBP, 5, 40, 72, 10), b2 = run(DDPMIR_ARCH_AVIF, 3, 32, 64, 10);
  std::printf("webp %08x %08x avif %08x %08x\n", a, a2, b, b2);
  return (a == a2 && b == b2 && a != 0 && b != 0) ? 0 : 1;
}

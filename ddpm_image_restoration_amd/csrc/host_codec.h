// Built-in host codec pool for the families without a device codec (host_codec.cpp).  Plain C++ (no HIP): the file also
// builds with gcc + AddressSanitizer on the CPU (tests/test_host_codec.py::test_sanitizer_build).
#pragma once
#include <string>

#include "../../include/ddpmir.h"

namespace ddpmir {

void set_codec_library(int arch, const std::string& path);
// NULL when the codec library cannot be loaded or fails its self-check, with the reason in builtin_host_codec_error()
ddpmir_codec_fn builtin_host_codec(int arch);
std::string builtin_host_codec_error();
// threads of this process' codec pool: CPUs this process may run on / LOCAL_WORLD_SIZE, at most 32 (DDPMIR_CODEC_THREADS overrides)
int builtin_host_codec_threads();
// the built-in pool with a per-image status: failed[i] != 0 marks an image the codec could not round-trip (nothing written for
// it); returns 0 unless the codec library itself is unavailable (-1)
int builtin_host_codec_status(int arch, const uint8_t* in, uint8_t* out, int B, int H, int W, int quality, uint8_t* failed);

}  // namespace ddpmir

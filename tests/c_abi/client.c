/* Plain-C client of libddpmir.so: no Python, no torch, nothing but include/ddpmir.h and the HIP runtime's C API for
 * device buffers.  Mirrors what a maintainer's binding does around webp_inference.py:618-635,682: create the model,
 * load the state dict tensor by tensor, run one forward (webp_inference.py:577) and a few DDRM steps with the on-device
 * JPEG codec, and write both results to a file that tests/test_c_abi_client.py compares with the Python binding.
 *
 *   client <weights.bin> <input.bin> <output.bin> [family: 0 WebP | 1 JPEG (default) | 2 AVIF]
 * The sampling loop runs with codec == NULL: the on-device JPEG codec, or for WebP / AVIF the library's built-in host codec
 * pool (libwebp / libavif located through $DDPMIR_LIBWEBP / $DDPMIR_LIBAVIF or the system's copies) -- no callback, so a
 * plain-C caller needs nothing but this header.
 * weights.bin: repeated { int32 name_len; char name[]; int32 ndim; int64 shape[ndim]; float data[prod(shape)] }
 * input.bin:   int32 B, H, W, quality, steps; float x[B*3*H*W]; float t[B]; float cl[B]
 * output.bin:  float x_theta[B*3*H*W]; float x_restored[B*3*H*W]
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ddpmir.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)r_, __LINE__); return 3; } } while (0)
#define CHECK_RC(ctx, e) do { int r_ = (e); if (r_ != DDPMIR_OK) { fprintf(stderr, "ddpmir error %d at line %d: %s\n", r_, __LINE__, ddpmir_last_error(ctx)); return 4; } } while (0)

static int read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 4 && argc != 5) { fprintf(stderr, "usage: client weights.bin input.bin output.bin [family]\n"); return 2; }
  ddpmir_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.arch = argc == 5 ? atoi(argv[4]) : DDPMIR_ARCH_JPEG; cfg.dtype = DDPMIR_F32; cfg.attn_min_level = 0; cfg.device = 0; cfg.use_graph = 0;
  ddpmir_ctx* ctx = ddpmir_create(&cfg);
  if (!ctx) { fprintf(stderr, "create failed: %s\n", ddpmir_last_error(NULL)); return 4; }

  FILE* fw = fopen(argv[1], "rb");
  if (!fw) { perror("weights"); return 2; }
  long n_tensors = 0;
  for (;;) {
    int32_t name_len, ndim;
    if (!read_exact(fw, &name_len, 4)) break;
    char name[256];
    int64_t shape[8], numel = 1;
    if (name_len <= 0 || name_len >= (int)sizeof name || !read_exact(fw, name, (size_t)name_len) || !read_exact(fw, &ndim, 4) ||
        ndim < 0 || ndim > 8 || !read_exact(fw, shape, 8 * (size_t)ndim)) { fprintf(stderr, "corrupt weights file\n"); return 2; }
    name[name_len] = 0;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    float* data = (float*)malloc((size_t)numel * 4 + 4);
    if (!data || !read_exact(fw, data, (size_t)numel * 4)) { fprintf(stderr, "short read on %s\n", name); return 2; }
    CHECK_RC(ctx, ddpmir_load_tensor(ctx, name, data, DDPMIR_F32, shape, ndim));
    free(data);
    ++n_tensors;
  }
  fclose(fw);
  CHECK_RC(ctx, ddpmir_finalize_weights(ctx, NULL, 1));
  printf("tensors %ld params %lld version %s\n", n_tensors, (long long)ddpmir_param_count(ctx), ddpmir_version());

  FILE* fi = fopen(argv[2], "rb");
  if (!fi) { perror("input"); return 2; }
  int32_t hdr[5];
  if (!read_exact(fi, hdr, sizeof hdr)) return 2;
  const int B = hdr[0], H = hdr[1], W = hdr[2], quality = hdr[3], steps = hdr[4];
  const size_t n = (size_t)B * 3 * H * W;
  float* hx = (float*)malloc(n * 4); float* ht = (float*)malloc((size_t)B * 4); float* hc = (float*)malloc((size_t)B * 4);
  if (!read_exact(fi, hx, n * 4) || !read_exact(fi, ht, (size_t)B * 4) || !read_exact(fi, hc, (size_t)B * 4)) return 2;
  fclose(fi);

  float *dx, *dt, *dc, *dy, *dr;
  CHECK_HIP(hipSetDevice(0));
  CHECK_HIP(hipMalloc((void**)&dx, n * 4)); CHECK_HIP(hipMalloc((void**)&dy, n * 4)); CHECK_HIP(hipMalloc((void**)&dr, n * 4));
  CHECK_HIP(hipMalloc((void**)&dt, (size_t)B * 4)); CHECK_HIP(hipMalloc((void**)&dc, (size_t)B * 4));
  CHECK_HIP(hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dt, ht, (size_t)B * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dc, hc, (size_t)B * 4, hipMemcpyHostToDevice));

  CHECK_RC(ctx, ddpmir_predict(ctx, dx, dt, dc, dy, B, H, W, NULL));                 /* model(x_t, t, compression_level) */
  ddpmir_loop_opts o;
  memset(&o, 0, sizeof o);
  o.seed = 7; o.noise_coeff = -1; o.phase_q = -1; o.phase_every = -1; o.phase_alpha = -1; o.start_step = -1; o.num_steps = 0;
  CHECK_RC(ctx, ddpmir_ddrm_sample_loop(ctx, dx, dr, B, H, W, quality, steps, 0.85, 1.0, &o));   /* sampler.sample(...) */
  CHECK_HIP(hipDeviceSynchronize());

  float* out = (float*)malloc(2 * n * 4);
  CHECK_HIP(hipMemcpy(out, dy, n * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out + n, dr, n * 4, hipMemcpyDeviceToHost));
  FILE* fo = fopen(argv[3], "wb");
  if (!fo || fwrite(out, 4, 2 * n, fo) != 2 * n) { perror("output"); return 2; }
  fclose(fo);

  /* error behaviour: bad sizes are reported, never abort */
  if (ddpmir_predict(ctx, dx, dt, dc, dy, B, 33, W, NULL) == DDPMIR_OK) { fprintf(stderr, "H=33 was accepted\n"); return 5; }
  printf("expected error: %s\n", ddpmir_last_error(ctx));
  ddpmir_destroy(ctx);
  printf("ok\n");
  return 0;
}

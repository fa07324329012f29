/*
 * ddpmir.h -- C ABI of the MI355X-native reverse-diffusion restoration sampler (libddpmir.so).
 *
 * The reference (Azure0413/DDPM_Image_Restoration) has no FFI layer; its boundary is two duck-typed Python
 * objects and one file format.  Every entry point below names the reference interface it
 * replaces.  All entry points return 0 on success or a negative DDPMIR_E* code; the message is available
 * from ddpmir_last_error().  No torch types cross this boundary: plain pointers and sizes only.
 *
 * Pointer convention: image / tensor arguments are DEVICE pointers on the context's HIP device unless the
 * name ends in _host.  Images are float32 NCHW in [-1, 1], exactly what the reference passes around.
 * A context is bound to one GPU and is not thread-safe; different contexts may run concurrently.
 */
#ifndef DDPMIR_H
#define DDPMIR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDPMIR_OK 0
#define DDPMIR_EINVAL (-1)   /* bad argument / unknown tensor name / shape mismatch */
#define DDPMIR_ESTATE (-2)   /* call order violated (e.g. predict before finalize_weights) */
#define DDPMIR_EHIP (-3)     /* HIP runtime error */
#define DDPMIR_ECODEC (-4)   /* codec callback failed or missing */
#define DDPMIR_ENOMEM (-5)

/* model families: webp_inference.py:330 WebPDiffusionModel, svd.ipynb c0:297 JPEGDiffusionModel,
 * avif_inference.py:316 AVIFDiffusionModel */
#define DDPMIR_ARCH_WEBP 0
#define DDPMIR_ARCH_JPEG 1
#define DDPMIR_ARCH_AVIF 2

#define DDPMIR_F32 0
#define DDPMIR_BF16 1
#define DDPMIR_F16 2         /* storage type of checkpoint tensors only (ddpmir_load_tensor) */
#define DDPMIR_FP8 3         /* arithmetic mode: DDPMIR_BF16 with OCP e4m3 operands (per-tensor power-of-two weight scale, unit
                                activation scale) for the 3x3 convolutions with >= 256 input and output channels, on
                                v_mfma_scale_f32_32x32x64_f8f6f4 -- the "fp8 MFMA deep convs" of BASELINE configs[4] */

typedef struct ddpmir_ctx ddpmir_ctx;

typedef struct ddpmir_config {
  int arch;            /* DDPMIR_ARCH_* */
  int dtype;           /* DDPMIR_F32 (parity mode) | DDPMIR_BF16 (bf16 matrix-core operands, fp32 accumulation and storage)
                          | DDPMIR_FP8 (bf16 mode + e4m3 operands in the deep 3x3 convolutions) */
  int attn_min_level;  /* 0 = self-attention in all 13 blocks (the reference, webp_inference.py:317-321);
                          5 = bottleneck blocks only (the variant BASELINE.json's configs describe) */
  int device;          /* HIP device ordinal */
  int use_graph;       /* 1 = replay captured hipGraphs: the U-Net forward in ddpmir_predict, and in ddpmir_ddrm_sample_loop the
                          whole timestep {t, forward, quantise, codec or D2H | H2D, noise, DDRM update, phase fix} with the
                          per-step scalars in a device-side parameter block (bit-identical to the eager launches) */
} ddpmir_config;

/* replaces: WebPDiffusionModel().to(device)  (webp_inference.py:618) */
ddpmir_ctx* ddpmir_create(const ddpmir_config* cfg);
void ddpmir_destroy(ddpmir_ctx* ctx);
const char* ddpmir_last_error(ddpmir_ctx* ctx);   /* ctx may be NULL: error of the last failed create */

/* replaces: model.load_state_dict(ckpt['model_state_dict'])  (webp_inference.py:621-627).
 * Call once per state_dict entry (torch does the unpickling), then finalize.  data_host: C order, element type `dtype`
 * = DDPMIR_F32 | DDPMIR_BF16 | DDPMIR_F16 (checkpoints saved in half precision are widened to fp32 exactly). */
int ddpmir_load_tensor(ddpmir_ctx* ctx, const char* name, const void* data_host, int dtype, const int64_t* shape, int ndim);
/* repack (OIHW -> [tap][Cout][Cin]) and upload; errors list missing names.  If dev_blob != NULL the packed
 * weights live in that caller-owned device buffer of ddpmir_weight_blob_bytes() bytes (fill=1: upload into it;
 * fill=0: adopt contents already there, e.g. after an RCCL broadcast from rank 0). */
size_t ddpmir_weight_blob_bytes(ddpmir_ctx* ctx);
int ddpmir_finalize_weights(ddpmir_ctx* ctx, void* dev_blob, int fill);
int ddpmir_weight_layout_version(void);            /* bumped whenever the packed blob layout changes (checked by converters) */
int64_t ddpmir_param_count(ddpmir_ctx* ctx);       /* sum(p.numel()) -- the harness prints it */

/* replaces: x_theta = model(x_t, t, compression_level)  (webp_inference.py:577; north_star's predict_noise()).
 * x, out: [B,3,H,W]; t, cl: [B].  H and W must be multiples of 32 (five 2x2 pools). Asynchronous on `stream`
 * (a hipStream_t; NULL = the context's own stream, synchronised before return). */
int ddpmir_predict(ddpmir_ctx* ctx, const float* x, const float* t, const float* cl, float* out, int B, int H,
                   int W, void* stream);

/* codec round trip on uint8 NCHW images, host memory: out = decode(encode(in, quality)).
 * replaces the PIL calls in webp_compress / avif_compress / jpeg_compress (webp_inference.py:512-525). */
typedef int (*ddpmir_codec_fn)(void* user, const uint8_t* in_u8_nchw, uint8_t* out_u8_nchw, int B, int H, int W,
                               int quality);

/* Built-in codecs (what `codec == NULL` selects in the sampling loop): JPEG = the bit-exact on-device libjpeg-turbo
 * simulator; WebP / AVIF = a C++ thread pool calling libwebp / libavif through their C APIs with the settings Pillow uses
 * (bit-identical to the reference's PIL calls with the same library build).  The libraries are opened at run time:
 * ddpmir_set_codec_library(DDPMIR_ARCH_WEBP | DDPMIR_ARCH_AVIF, path) names the file (the Python host passes the copy
 * bundled with its Pillow); otherwise $DDPMIR_LIBWEBP / $DDPMIR_LIBAVIF, then the system's libwebp.so.7 / libavif.so.16. */
int ddpmir_set_codec_library(int arch, const char* path);
/* out = decode(encode(in, quality)) with the built-in HOST codec of a family, uint8 NCHW in host memory (no GPU, no
 * context): replaces webp_compress / avif_compress when degrading inputs (webp_inference.py:669). */
int ddpmir_host_codec_roundtrip_u8(int arch, const uint8_t* in_u8_nchw_host, uint8_t* out_u8_nchw_host, int B, int H, int W,
                                   int quality);
/* threads of this process' built-in codec pool: the CPUs it may run on divided by LOCAL_WORLD_SIZE (one rank per GPU: the
 * ranks of a host share its cores), at most 32; $DDPMIR_CODEC_THREADS overrides */
int ddpmir_host_codec_threads(void);

typedef struct ddpmir_loop_opts {
  ddpmir_codec_fn codec;      /* NULL = the family's built-in codec (see above) */
  void* codec_user;
  const float* noise;         /* optional device pointer [steps][B][3][H][W]: noise[i] is drawn at loop index i
                                 (replaces torch.randn_like, webp_inference.py:589); NULL = Philox(seed) */
  uint64_t seed;
  uint64_t image_offset;      /* global index of image 0 of this shard (multi-GPU: the noise does not depend on the number of ranks;
                                 fp32 mode is bit-identical for any sharding, bf16 / fp8 mode within its noise floor -- kernel
                                 selection follows the per-rank batch) */
  float* trace;               /* optional device pointer [steps][2][B][3][H][W]: x_theta, x_t after each step */
  double noise_coeff;         /* <0 = the model family's default (0.2 / 0.15) */
  int phase_q;                /* <0 = default quality threshold of the phase fix (15 / 20 / 30) */
  int phase_every;            /* <=0 = default (5 / 3) */
  double phase_alpha;         /* <0 = default (0.7 / 0.8) */
  int start_step;             /* <0 = steps-1.  Run the loop from index i = start_step ... */
  int num_steps;              /* <=0 = down to i = 0; else only this many iterations (bench.py times a slice of the
                                 T-step schedule; x_t is then NOT a finished restoration) */
} ddpmir_loop_opts;

/* replaces: DDRM{WebP,JPEG,AVIF}Sampler(model).sample(x_t, quality, steps, eta, eta_b)
 * (webp_inference.py:557-602, avif_inference.py:414-459, svd.ipynb c0:395-440).  Blocking.
 * eta / eta_b / alpha are doubles because the reference forms (1 - eta_b) and (1 - alpha) in Python double
 * precision before the fp32 tensor multiply. */
int ddpmir_ddrm_sample_loop(ddpmir_ctx* ctx, const float* y, float* out, int B, int H, int W, int quality,
                            int steps, double eta, double eta_b, const ddpmir_loop_opts* opts);

/* eps-predict DDPM flavour.  replaces: ddpm_inference(model, x0, T)  (experiments/code/conv.ipynb c1:117-125)
 * on a context created with ddpmir_create_eps_unet(); x in [0,1], betas_host: float32[T_total].  Blocking. */
ddpmir_ctx* ddpmir_create_eps_unet(int device);
int ddpmir_predict_noise(ddpmir_ctx* ctx, const float* x, float t, float* eps, int B, int H, int W);
int ddpmir_p_sample_loop(ddpmir_ctx* ctx, const float* x, float* out, int B, int H, int W, int T,
                         const float* betas_host, int n_betas);

/* standalone pieces, exported for the harness and for parity tests */
int ddpmir_phase_consistency(ddpmir_ctx* ctx, const float* x, const float* ref, float* out, int B, int H, int W,
                             double alpha);
int ddpmir_jpeg_roundtrip_u8(ddpmir_ctx* ctx, const uint8_t* in_u8_nchw, uint8_t* out_u8_nchw, int B, int H, int W,
                             int quality);   /* device pointers; bit-exact libjpeg-turbo simulator */

/* HIP-event timing of kernel classes on the context's stream (bench.py's roofline block).  enable=1 brackets
 * every launch of the profiled classes with events; read returns the accumulated device milliseconds, launch
 * count and algorithmic FLOPs since the last enable.  Classes: "attention_l0" (self-attention over all H*W
 * tokens of the full-resolution blocks), "attention" (all 13; needs enable = 2), "forward" (whole U-Net forwards),
 * "codec_host" (HOST milliseconds spent inside the codec callback / built-in pool of the sampling loop, and the number of
 * calls).  enable = 1 brackets the forwards and the full-resolution attention launches only (6 event records per
 * forward); enable = 2 every attention launch (54 records per forward); 0 stops. */
/* enable = 3 additionally brackets every GEMM launch (conv3x3 / 1x1 / projections of the DMA-streamed bf16 / fp8 path): classes
 * "gemm" and "gemm_fp8" with algorithmic FLOPs 2*M*N*K -- about 220 event records per forward, so not for timed regions.
 * "attention_general": launches = workgroups of the general (running-max) attention kernel that had work to do, flops = such
 * workgroups launched beside a no-max kernel (the share of query blocks whose score bound forced the running max). */
int ddpmir_profile_enable(ddpmir_ctx* ctx, int enable);
int ddpmir_profile_read(ddpmir_ctx* ctx, const char* cls, double* ms, int64_t* launches, double* flops);
/* "ddpmir <semver> (gfx950) src=<sha256/16 of csrc + include> hipcc=<version the library was built with>": lets a caller
 * (bench.py prints it) tell a stale prebuilt binary from a fresh compile of the sources beside it. */
const char* ddpmir_version(void);

/* Tuning / diagnostic options of a context.  "attention_no_max_bound" (default 80, log2 units): a block of queries runs in
 * the attention kernel WITHOUT a running softmax maximum when max|q| of the block * max|k| of its (batch, head) pair stays
 * below it (scores then cannot leave fp32's range); a negative value forces the general kernel everywhere (the worst case
 * bench.py reports as general_kernel_only).
 * "fp8_min_channels" (64 / 128 / 256, default 256; DDPMIR_FP8, before the weights are loaded): the 3x3 convs with at least that
 * many input and output channels take e4m3 operands.
 * "bf16_activation_stream" (-1 / 0 / 1): the tensors BETWEEN residual blocks (block outputs, pooled / upsampled tensors, concat
 * buffers) are stored as bf16 instead of fp32 -- the rounding the reference's own autocast applies to every conv output.
 * -1 (default): in DDPMIR_FP8 mode only, whose error budget e4m3 dominates; 1: in DDPMIR_BF16 mode too (measured: forward error
 * against the fp32 oracle mean 2.0e-3 instead of 1.7e-3 at 256 x 256; restored PSNR against fp32 mode over 64 images
 * +0.0031 +- 0.0018 dB instead of +0.0034 +- 0.0016, i.e. unchanged; -7 % time at 512 x 512); 0: never.  AVIF and DDPMIR_F32
 * always keep fp32.
 * "gemm256_tile" (-1 / 0 / 1, bf16 / fp8 modes): the 3x3 convs with >= 256 output channels on the 256 x 256 eight-wave tile
 * (csrc/gemm256.hip) -- -1 (default): where its tiles (x K splits) fill the chip; 0: never (the 128-row kernel, for A/B runs and
 * parity tests); 1: wherever the shape is supported.
 * "epilogue_direct" (1 default / 0): the GEMM kernels with <= 64 output columns store their accumulators straight from registers
 * (no LDS-transposed tile); 0 keeps the LDS epilogue everywhere (A/B runs; the results are bit-identical, the fused GroupNorm sums
 * equal up to their summation order).
 * "conv_tile" (0 / 1 default / 2): GroupNorm-apply (+ GELU) fused into the full-resolution 64 -> 64 3x3 convs (conv_tile.hip: the raw
 * tensor is read once, no bf16 operand pass) -- 1: from 1.5 M pixels per launch (where it measured faster), 2: wherever supported,
 * 0: operand pass + DMA GEMM.
 * "head_fused" (1 default / 0): the output head (GroupNorm-apply + SiLU + conv 64 -> 3 + tanh + NCHW) as one kernel; 0: three launches.
 * "concurrent_lanes" (-1 / 0 off, default; 1; 2): the device-codec JPEG loop in bf16 / fp8 mode splits the batch into two lanes
 * that run concurrently on two streams (a second context on the same weight blob); 2 additionally makes the full-resolution
 * attention launches of the lanes take turns.  Each lane's result equals the single-lane loop's on its half of the batch.  An
 * experiment that measured +0.5-1 % on 256 x 256, B = 16 (+2 % at 128 x 128, B = 64): off by default. */
int ddpmir_set_option(ddpmir_ctx* ctx, const char* name, double value);

#ifdef __cplusplus
}
#endif
#endif /* DDPMIR_H */

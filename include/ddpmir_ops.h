/*
 * ddpmir_ops.h -- single-operator entry points of libddpmir.so on HOST buffers (the library stages them
 * through the GPU).  They exist so every row of the hot-path table can be checked
 * against the oracle in isolation; the product path is ddpmir.h.  All tensors fp32, NHWC unless noted.
 * Each returns 0 or a negative DDPMIR_E* code; message via ddpmir_ops_last_error().
 */
#ifndef DDPMIR_OPS_H
#define DDPMIR_OPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* ddpmir_ops_last_error(void);

/* nn.Conv2d 3x3 pad 1 / 1x1 / nn.Linear as implicit GEMM (webp_inference.py:282,292,229,215).
 * x: [B,H,W,Cin] NHWC; w_oihw: torch layout [Cout][Cin][k][k]; bias [Cout] or NULL.
 * Optional prologue: per-(b,c) scale/shift [B][Cin] then pro_act (0 none,1 gelu,2 silu), pro_mask (0,1 low,2 high).
 * Optional epilogue: tb [B][Cout]; epi_act (0..6); mul [B,H,W,Cout]; rowscale [B]; epi_mask; res [B,H,W,Cout].
 * out: [B,H,W,Cout] NHWC or [B,Cout,H,W] if nchw_out. */
int ddpmir_op_conv(const float* x, int B, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout,
                   int ksize, const float* scale, const float* shift, int pro_act, int pro_mask, int bs, int low,
                   const float* tb, int epi_act, const float* mul, const float* rowscale, int epi_mask,
                   const float* res, int nchw_out, float* out);
/* select the bf16 matrix-core variant for subsequent ddpmir_op_conv calls of this thread (0 = fp32) */
void ddpmir_op_set_bf16(int enable);
/* gate-select epilogue of the DMA GEMM for subsequent ddpmir_op_conv calls (0 = off): Cout = 2C; per pixel only the
 * low-frequency half [0,C) or the high-frequency half [C,2C) is kept (webp_inference.py:241-252 masks), so mul / res /
 * out are [B,H,W,C] and rowscale multiplies the high half only */
void ddpmir_op_set_gate_select(int C);
/* IGemm::tile256 of the GEMM ops: -1 = the 256 x 256 eight-wave tile where it fills the chip (default), 0 = never, 1 = wherever supported */
void ddpmir_op_set_tile256(int mode);
/* IGemm::epi_direct_opt of the GEMM ops: 1 (default) = the register-direct epilogue of the 64-column kernels where it applies, 0 = the LDS-transposed one */
void ddpmir_op_set_epilogue_direct(int on);
/* launches that tile has taken in this process (proof for a test that its shape did not fall back to the 128-row kernel) */
long long ddpmir_op_gemm256_launches(void);
/* device-resident timing of one conv/linear launch on random data (ms per call); gelu_norm adds the fused
 * GroupNorm+GELU prologue; dbg is a kernel-internal experiment mask (0 in normal use) */
int ddpmir_op_conv_time(int B, int H, int W, int Cin, int Cout, int ksize, int bf16, int gelu_norm, int residual,
                        int dbg, int iters, float* ms_per_call);

/* nn.GroupNorm statistics folded to per-(b,c) scale/shift (webp_inference.py:281); x [B,HW,C]. */
int ddpmir_op_gn_scale_shift(const float* x, int B, int HW, int C, int G, const float* gamma, const float* beta,
                             float* scale, float* shift);

/* attention core on qkv [B,N,3C] -> out [B,N,C] (webp_inference.py:319) */
int ddpmir_op_attention(const float* qkv, int B, int N, int C, int heads, float* out);
/* same through the bf16 matrix-core kernel (fp32 in/out, bf16 operands, fp32 accumulation) */
int ddpmir_op_attention_bf16(const float* qkv, int B, int N, int C, int heads, float* out);
/* the same with the routing bound of the no-max kernel (AttnOpts::nm_bound; < 0 = general kernel only) and the routing
 * counters: stats[0] = general-kernel workgroups that had work, stats[1] = general-kernel workgroups launched beside a
 * no-max kernel (0 when the shape has no no-max variant) */
int ddpmir_op_attention_bf16_ex(const float* qkv, int B, int N, int C, int heads, float nm_bound, float* out, int64_t* stats);
/* conv through the DMA GEMM whose epilogue also accumulates the GroupNorm statistics of its OUTPUT (the next layer's
 * nn.GroupNorm, webp_inference.py:290): sums [B][groups][2] = (sum, sum of squares) per sample and group. */
int ddpmir_op_conv_gnstats_bf16(const float* x, int B, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout,
                                int ksize, int groups, float* out, double* sums);
/* conv_tile.hip: out = conv3x3(act(x * scale[b][c] + shift[b][c])) + bias (+ tb[b][n]) with the GroupNorm-apply pass fused into the
 * conv (webp_inference.py:306-313).  x [B,H,W,Cin] (rounded to bf16 first when in16), Cout = 64, Cin % 64 == 0, H % 8 == 0, W % 32 == 0;
 * scale / shift / tb may be NULL; groups > 0: also the per-(sample, group) sums of the stored output [B][groups][2]; out (fp32) and / or
 * out16f (the bf16 copy, widened) ; raw16f != NULL: the bf16 copy of x the kernel writes for the 1x1 shortcut, widened. */
int ddpmir_op_conv_tile(const float* x, int B, int H, int W, int Cin, int in16, const float* scale, const float* shift, int act,
                        const float* w_oihw, const float* bias, const float* tb, int Cout, int groups, int want16, float* out, float* out16f,
                        double* sums, float* raw16f);
/* out = conv_kxk(x1, w1) + conv_1x1(x2, w2) + bias in ONE DMA-GEMM launch (second K segment): conv_out with the block's
 * 1x1 shortcut folded in (webp_inference.py:326-327).  x1 [B,H,W,C1], x2 [B,H,W,C2], out [B,H,W,Cout]. */
int ddpmir_op_conv_dual_bf16(const float* x1, int B, int H, int W, int C1, const float* w1_oihw, int ksize, const float* x2,
                             int C2, const float* w2_oi, const float* bias, int Cout, float* out);
/* the same in DDPMIR_FP8 arithmetic: the 3x3 segment with OCP e4m3 operands (per-tensor power-of-two weight scale), the 1x1
 * shortcut segment in bf16 with its weights on the same scale, one acc_scale in the epilogue (Context::conv8 with a shortcut). */
int ddpmir_op_conv_dual_fp8(const float* x1, int B, int H, int W, int C1, const float* w1_oihw, const float* x2, int C2,
                            const float* w2_oi, const float* bias, int Cout, float* out);
/* qkv projection (nn.MultiheadAttention in_proj, webp_inference.py:295) + attention core through the bf16 path:
 * x [B,N,C] -> out [B,N,C] (before out_proj).  fused = 1: the projection's epilogue writes the attention operand
 * images directly (head_dim 16, 4 heads, N % 32 == 0); fused = 0: fp32 qkv tensor + separate packing kernel. */
int ddpmir_op_qkv_attention_bf16(const float* x, int B, int N, int C, int heads, const float* w_in, const float* b_in,
                                 int fused, float* out);
/* shader clock under load: a one-wave kernel on its own stream compares s_memtime with s_memrealtime for `ms` of wall time while
 * the caller keeps the device busy from other streams; _end waits for it and returns the effective clock in GHz */
int ddpmir_op_clock_sample_begin(double ms);
int ddpmir_op_clock_sample_end(double* ghz, double* ms);
/* device-resident timing of the attention kernels on random data: mode 0 = fp32, 1 = bf16; average ms per call */
int ddpmir_op_attention_time(int B, int N, int C, int heads, int mode, int iters, float* ms_per_call);

/* blockwise T X T^T (DCTLayer webp_inference.py:161-192 / AVIFAdaptiveTransform avif_inference.py:140-177) */
int ddpmir_op_block_transform(const float* x, int B, int H, int W, int C, int bs, const float* mats, int per_channel,
                              float mix_in, float mix_t, float* out);

/* frequency block of the DCT families in bf16 mode (webp_inference.py:231-270): S = x + gate(DCT(x)) * DCT(x) as bf16 values;
 * w1 [C][C] = both gates' first layers stacked, w2 [2C][C] block-diagonal second layers (model.hip put_gate_pair);
 * fused = 0: DCT kernel + two DMA GEMMs, fused = 1: the one-pass kernel (C = 64 or 128) */
int ddpmir_op_freq_gates_bf16(const float* x, int B, int H, int W, int C, int bs, int low, const float* dct, const float* w1,
                              const float* b1, const float* w2, const float* b2, const float* boost, int fused, float* out,
                              const float* att /* or NULL: x := wo att + bo + x first (attention out_proj + residual) */,
                              const float* wo, const float* bo);
int ddpmir_op_maxpool2(const float* x, int B, int H, int W, int C, float* out);        /* webp_inference.py:342 */
int ddpmir_op_upsample2(const float* x, int B, int H, int W, int C, float* out);       /* webp_inference.py:389 */
int ddpmir_op_bilinear(const float* x, int B, int Hi, int Wi, int C, int Ho, int Wo, float* out);
/* The passes between residual blocks in bf16 / fp8 mode: mode 0 statistics alone, 1 max-pool 2x2 (webp_inference.py:342),
 * 2 bilinear x2 (:389) on NHWC tensors; with G > 0 the pass also leaves GroupNorm partial sums of what it produced, returned
 * here folded into scale / shift [B][C] (webp_inference.py:277-279).  f32 = 0: the bf16 activation stream (x is rounded to bf16
 * on the way in, rows ldx >= C elements apart; out = the bf16 result widened to fp32 -- the rounding the reference's autocast
 * applies to conv outputs); f32 = 1: fp32 tensors (modes 1, 2) */
int ddpmir_op_stream16(int mode, const float* x, int B, int H, int W, int C, int ldx, int G, const float* gamma, const float* beta,
                       float* out, float* scale, float* shift, int f32);
int ddpmir_op_adaptive_avgpool(const float* x, int B, int H, int W, int C, int S, float* out);

/* TimeEmbedding.forward (webp_inference.py:145-151): t [B] -> temb [B,256] */
int ddpmir_op_time_embed(const float* t, int B, const float* w0, const float* b0, const float* w2, const float* b2,
                         float* temb);

/* uint8 quantisation (webp_inference.py:509) and one DDRM update (webp_inference.py:584-600) on NCHW images */
int ddpmir_op_quantize_u8(const float* x, int64_t n, uint8_t* out);
int ddpmir_op_ddrm_update(const float* x_theta, const uint8_t* dec_u8, const float* y, const float* noise,
                          int64_t n_per_image, int B, double eta, double eta_b, double noise_coeff, int step_i, int steps,
                          float* out);
int ddpmir_op_randn(int64_t n, uint64_t seed, uint64_t offset, float* out);

#ifdef __cplusplus
}
#endif
#endif

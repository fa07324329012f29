// Launch wrappers of every HIP kernel on the restoration hot path (gfx950).  All tensors are fp32 NHWC
// with an explicit pixel-row stride (ld, in floats) so a block can write straight into a channel slice of
// a concat buffer.  Every launcher is asynchronous on `stream` and allocation-free (graph-capturable).
#pragma once
#include "common.h"

namespace ddpmir {

// ------------------------------------------------------------------------------------------------ igemm
// out[m][n] = epilogue( sum_{tap,c} prologue(A[pix(m)+tap][c]) * Wt[tap][n][c] )
// Covers nn.Conv2d 3x3 pad 1 (taps=9), 1x1 conv / nn.Linear (taps=1).  Reference: a3/a4/a6/a8 of SURVEY 8(a).
// The zero page the DMA GEMM reads halo / tail chunks from.  64 KiB, not 16 B: every lane of every workgroup reads its
// own 16-byte slot, so border-heavy shapes (8x8 maps: 30 % of the A loads) do not all hammer one L2 channel.
constexpr int GEMM16_ZERO_BYTES = 65536;
constexpr size_t GEMM16_SPLITK_WS_BYTES = size_t(64) << 20;   // what the model / op wrappers reserve for split-K partials

struct IGemm {
  // input
  const float* A = nullptr; int lda = 0;      // [B*H*W][lda], channels 0..Cin-1 used (Cin % 4 == 0)
  int B = 1, H = 1, W = 1, Cin = 0;
  // weights, packed [taps][Cout][Cin]
  const float* Wt = nullptr; int Cout = 0; int taps = 1;
  const void* Wt16 = nullptr;                 // same layout in bf16: selects the bf16 matrix-core kernel
  const float* bias = nullptr;                // [Cout] or null
  // prologue: v = A*scale[b][c] + shift[b][c]; act; static frequency mask on the pixel position
  const float* scale = nullptr; const float* shift = nullptr;   // [B][Cin] or null
  int pro_act = ACT_NONE;
  int pro_mask = MASK_NONE; int bs = 4, low = 3;
  // epilogue: v = act(acc + bias + tb[b][n]); v *= mul[m][n]; v *= rowscale[b]; v = mask? v:0; v += res[m][n]
  const float* tb = nullptr; int ldtb = 0;
  int epi_act = ACT_NONE;
  const float* mul = nullptr; int ldmul = 0;
  const float* rowscale = nullptr;            // [B]
  int epi_mask = MASK_NONE;
  const float* res = nullptr; int ldres = 0;
  const void* res16 = nullptr;                // gemm16.hip: the residual as a bf16 tensor instead (same ldres; bf16 activation stream)
  // output
  float* out = nullptr; int ldo = 0; int store = STORE_NHWC;
  // DMA-streamed bf16 path (gemm16.hip): A already holds what the GEMM consumes, as bf16 [B*H*W][lda16]
  const void* A16 = nullptr; int lda16 = 0;
  const void* zero16 = nullptr;               // GEMM16_ZERO_BYTES of zeros in device memory (halo / tail source)
  void* out16 = nullptr; int ldo16 = 0;       // optional bf16 copy of the epilogue result (out may then be null)
  // qkv projection fused with the attention operand packing (4 heads of head_dim 16 or 32: gemm16.hip epilogue writes
  // the Q / K / V^T images of attention_bf16.hip directly instead of an fp32 [M][192] tensor); see AttnPack16
  // per-pixel gate selection (frequency block, gemm16.hip): Cout = 2*sel_C; columns [0, sel_C) are the low-frequency gate,
  // [sel_C, 2 sel_C) the high-frequency gate; a row only stores the half that matches its pixel class, at column n mod sel_C
  // (mul / res / out / out16 are indexed with that column; rowscale applies to the high half only)
  int sel_C = 0;
  // GroupNorm statistics of the OUTPUT tensor emitted by the epilogue (gemm16.hip, non-split launches): per (sample, 32-row
  // wave tile, group) the sum and the sum of squares (fp32 within the tile, stored as doubles), one writer per slot; layout
  // [B][H*W/32][gn_groups][2] = gn_sums_doubles() doubles, no clearing needed.  launch_gn_from_sums folds them in a fixed
  // order.  Saves the separate statistics pass over the tensor (gemm16_can_fuse_gn() says when it applies).
  double* gn_sums = nullptr; int gn_groups = 0;
  // second K segment (gemm16.hip): a 1x1 operand pair appended after the taps of the first, e.g. the block's 1x1
  // shortcut conv folded into conv_out (out = W_out * S + W_sc * x + (b_out + b_sc)): saves one write + read of `out`
  const void* A16b = nullptr; int lda16b = 0; const void* Wt16b = nullptr; int Cinb = 0;
  // gemm16.hip, set by the launcher: 3x3 conv on an 8-channel operand (the image conv): the nine taps x 8 channels are ONE
  // K axis of 72 (a 16-byte chunk = one tap), two K stages instead of nine mostly-zero ones
  int tap_pack = 0;
  void* pk_q = nullptr; void* pk_kv = nullptr; float* pk_kmax = nullptr; int pk_nt = 0; float pk_qscale = 0.f; int pk_D = 16; int pk_heads = 4;
  float* pk_qmax = nullptr;                   // [bh][tile] max |q|^2 of the tile's 32 tokens (the score bound of the no-max attention kernel)
  // split-K partial sums [nsplit][M][Cout] fp32 (gemm16.hip): written by the splits, folded in fixed order by a second
  // kernel together with bias / time bias / residual / bf16 copy -> bf16 mode is bit-reproducible (no float atomics)
  float* splitk_ws = nullptr; size_t splitk_ws_bytes = 0;
  // fp8 operands (gemm16.hip, F8): A16 / Wt16 then point to OCP e4m3 tensors (lda16 / Cin in elements = bytes), acc_scale
  // undoes the per-tensor weight scale; out8 / ldo8: optional e4m3 copy of the epilogue result (operand of the next fp8 conv)
  int f8 = 0; float acc_scale = 1.f; void* out8 = nullptr; int ldo8 = 0;
  int nmajor = 0;   // set by the launcher: blockIdx.x is the N tile (weight-bound shapes), see gemm16.hip
  // gemm256.hip (the 256 x 256 eight-wave tile for 3x3 convs with Cout >= 256): -1 = where it fills the chip (default),
  // 0 = never (A/B runs, parity tests against the 128-row kernel), 1 = wherever gemm256_supported() allows
  int tile256 = -1;
  // gemm_epilogue.h, 64-column kernels: register-direct epilogue for the plain option sets (epilogue_direct_mode()); epi_direct is the
  // mode the launcher computed (0 = LDS-transposed epilogue), epi_direct_opt = 0 forces the LDS epilogue (A/B runs, parity tests)
  int epi_direct = 0, epi_direct_opt = 1;
  FastDiv fd_hw, fd_w;   // set by launch_gemm16: exact division by H*W and by W (pixel index -> sample, row, column)
  int dbg = 0;   // timing experiments only: 1 skip A loads, 2 skip B loads, 4 skip MFMA, 8 skip stores, 16 skip LDS writes
};
void launch_igemm(const IGemm& p, hipStream_t stream);
// up to four problems of equal (Cout, Cin, taps) in one launch (bf16 path; falls back to a loop otherwise)
void launch_igemm_group(const IGemm* ps, int n, hipStream_t stream);
bool gemm16_supported(const IGemm& p);
void launch_gemm16(const IGemm& p, hipStream_t stream);
bool gemm256_supported(const IGemm& p);                    // gemm256.hip
bool launch_gemm256(const IGemm& p, hipStream_t stream);   // false: not taken (the caller launches the 128-row kernel)
long long gemm256_launch_count();                          // launches the tile has taken in this process
inline size_t gn_sums_doubles(int B, int HW, int groups) { return (size_t)B * (size_t)(HW / 32) * groups * 2; }
bool gemm16_can_fuse_gn(const IGemm& p, int groups);   // p as it will be launched (without gn_sums)
// fold the [B][HW/32][G][2] partials written by the epilogue above into per-(b,c) scale / shift of a C-channel tensor.
// scale / shift rows are ldsc floats apart; gamma / beta / scale / shift may point at a channel offset of a wider tensor
// (the skip half of an up block's concat input gets its statistics from the down block's conv_out this way);
// zero_to > C: channels [C, zero_to) of every row get scale = shift = 0 (padding channels of the operand pass)
void launch_gn_from_sums(const double* sums, int B, int HW, int C, int G, const float* gamma, const float* beta, float* scale,
                         float* shift, hipStream_t stream, int ldsc = 0, int zero_to = 0);
// fp32 -> bf16 with GroupNorm affine + activation applied once for the consuming conv; optional raw bf16 copy
// (zero/nzero: optionally clears a small fp64 accumulator, e.g. IGemm::gn_sums of the GEMM that consumes this operand)
// x16in != null: the input is bf16 [M][ldx] there (x unused)
void launch_gn_apply16(const float* x, int ldx, int B, int HW, int C, const float* scale, const float* shift, int act,
                       void* out16, int ld16, void* raw16, int ldraw, hipStream_t s, double* zero = nullptr, int nzero = 0,
                       void* out8 = nullptr, int ld8 = 0, const void* x16in = nullptr);

// ------------------------------------------------------------------------------------------------ attention
// softmax(Q K^T / sqrt(d)) V per (batch, head); qkv is [B][N][3C] (q | k | v, heads contiguous in C),
// out is [B][N][ldo] (first C columns).  Reference: nn.MultiheadAttention core, webp_inference.py:319.
void launch_attention(const float* qkv, float* out, int ldo, int B, int N, int C, int heads, hipStream_t stream);

// bf16 matrix-core variant (attention_bf16.hip): qkv/out stay fp32 in HBM, Q/K/V are repacked to bf16 operand
// images in `workspace` (attention_bf16_workspace() bytes), softmax statistics and accumulators are fp32.
bool attention_bf16_supported(int N, int C, int heads);
// Operand images of the head_dim-16 attention kernel inside `workspace`, for producers that write them directly
// (the fused qkv projection).  Tile = 32 keys: K image 1024 B + V^T image (2 x 2 slabs of 17 rows x 16 B).
constexpr int ATTN16_KT_BYTES = 1024, ATTN16_RV = 17, ATTN16_TILE_BYTES = ATTN16_KT_BYTES + 4 * ATTN16_RV * 16;
struct AttnPack16 { void* q; void* kv; float* kmax; int ntiles; float qscale; int D; int heads; float* qmax2; };
bool attention_bf16_prepack_supported(int N, int C, int heads);
AttnPack16 attention_bf16_pack_targets(int B, int N, int C, int heads, void* workspace);
size_t attention_bf16_workspace(int B, int N, int C, int heads);
// Routing between the no-max kernel and the general (running-max) kernel, per (batch, head, block of 128..512 queries):
// a block runs without a running max when  max|q| of its queries * max|k| of its pair <= nm_bound  (log2 units; 80 keeps
// 2^s, the sums over 2^18 keys and the PV accumulators inside fp32).  nm_bound < 0 sends everything to the general kernel.
// general_taken (device) / general_total (host): profiling counters -- general-kernel workgroups that did the work /
// general-kernel workgroups launched beside a no-max kernel.
struct AttnOpts {
  float nm_bound = 80.0f;
  unsigned* general_taken = nullptr;
  int64_t* general_total = nullptr;
};
// out16 != nullptr: the result is written as bf16 [B][N][C] there instead of fp32 `out`
// qkv == nullptr: the operand images in `workspace` were already written by the producer (AttnPack16)
void launch_attention_bf16(const float* qkv, float* out, int ldo, int B, int N, int C, int heads, void* workspace,
                           hipStream_t stream, void* out16 = nullptr, const AttnOpts* opts = nullptr);

// ------------------------------------------------------------------------------------------------ pointwise
// GroupNorm statistics -> per-(b,c) scale/shift so that norm(x) = x*scale + shift (eps 1e-5, biased var).
// `partial` is scratch of >= B*G*gn_splits(...)*2 doubles.
int gn_splits(int HW, int C);
// ldsc > 0: scale / shift rows are ldsc floats apart and x / gamma / beta / scale / shift may point at a channel offset of a
// wider tensor (statistics of one half of a concat input); zero_to as in launch_gn_from_sums (default: pad C up to 4)
void launch_gn_scale_shift(const float* x, int ldx, int B, int HW, int C, int G, const float* gamma,
                           const float* beta, double* partial, float* scale, float* shift, hipStream_t stream,
                           int ldsc = 0, int zero_to = -1);

void launch_maxpool2(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s);
void launch_upsample2(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s);
// generic bilinear resize (align_corners=False, coordinate scale = in/out) -- AVIF pooled gates
void launch_bilinear(const float* in, int ldi, float* out, int ldo, int B, int Hi, int Wi, int Ho, int Wo,
                     int C, float scale_h, float scale_w, int accumulate, float out_scale, hipStream_t s);
void launch_adaptive_avgpool(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, int S,
                             hipStream_t s);
// all four AdaptiveAvgPool2d scales {1,2,4,8} from one pass: bin sums [B][64][C], then folded per scale
bool pool8_supported(int H, int W, int C);
void launch_pool8_sums(const float* in, int ldi, float* sums, int B, int H, int W, int C, hipStream_t s);
void launch_pool_from8(const float* sums, float* out, int ldo, int B, int H, int W, int C, int S, hipStream_t s);

// blockwise Y = T X T^T (block bs in {4,8}); mats: one [bs][bs] matrix (per_channel=0) or [C][bs][bs].
// out = mix_in * x + mix_t * Y   (mix_in=0,mix_t=1: plain transform; head: 1, 0.1)
// low16/high16 (optional): bf16 copies of the result masked to the low / high in-block frequencies [B*H*W][C]
// DCT + both frequency gates + S = x + g * xd in one pass (freq_fused.hip; bf16 mode, C = 64, whole bs x bs blocks)
bool freq_gate_fused_supported(int B, int H, int W, int C, int bs);
// (att16 != null: x := W_o att16 + b_o + x is formed inside, i.e. the attention's out_proj + residual rides along)
void launch_freq_gate_fused(const float* x, const float* dct, const void* w1, const float* b1, const void* w2, const float* b2,
                            const float* boost, void* S16, int B, int H, int W, int C, int bs, int low, hipStream_t s,
                            const void* att16 = nullptr, const void* wo = nullptr, const float* bo = nullptr, void* S8 = nullptr,
                            const void* xin16 = nullptr);
// (xin16 != null: x is read as bf16 [B*H*W][C] from there instead of fp32 from `x`)
// (S8 != null: S is written as OCP e4m3 there instead of bf16 at S16 -- the operand of an fp8 conv_out)
// gn_partial_out (mixed 64-channel output, whole blocks, shared matrix: block_transform_gn_fusable): GroupNorm(8) partial sums
// of the output per (sample, block, group), [B][(H/bs)*(W/bs)][8][2] doubles, for launch_gn_from_partials
void launch_block_transform(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, int bs,
                            const float* mats, int per_channel, float mix_in, float mix_t, hipStream_t s,
                            void* low16 = nullptr, void* high16 = nullptr, int low = 0, double* gn_partial_out = nullptr, int io16 = 0);
// (io16: `in` / `out` are bf16 tensors -- the head's mixed transform in the bf16 activation stream)
bool block_transform_gn_fusable(int H, int W, int C, int bs);
// fold [B][splits][G][2] double partials (any producer) into per-(b,c) scale / shift; n_per_group = elements per (sample, group)
// (ldsc / zero_to as in launch_gn_scale_shift: one half of a concat input)
void launch_gn_from_partials(const double* partial, int splits, int B, double n_per_group, int C, int G, const float* gamma,
                             const float* beta, float* scale, float* shift, hipStream_t stream, int ldsc = 0, int zero_to = -1);
// Inter-block stream passes with GroupNorm statistics (pointwise.hip stream_k; bf16 / fp8 mode): NHWC tensors with 16-byte aligned
// rows, 16 bytes * 2^k of channels per pixel.  2x2 max-pool / bilinear x2 (torch semantics as the plain kernels) / statistics
// alone; with `partial` (>= B * G * 512 * 2 doubles; stream_stats_supported: G <= 8 groups of 16 bytes * 2^k channels) the
// GroupNorm partial sums of the produced tensor come along.  Return value: splits per sample for launch_gn_from_partials
// (0 without statistics).  *_16: the bf16 activation stream (results rounded to bf16); *_stats: the fp32 stream.
bool stream_stats_supported(int C, int G, int elem_bytes);
int launch_stats16(const void* x16, int ldx, int B, int H, int W, int C, int G, double* partial, hipStream_t s);
int launch_maxpool2_16(const void* in16, int ldi, void* out16, int ldo, int B, int H, int W, int C, hipStream_t s,
                       double* partial = nullptr, int G = 0);
int launch_upsample2_16(const void* in16, int ldi, void* out16, int ldo, int B, int H, int W, int C, hipStream_t s,
                        double* partial = nullptr, int G = 0);
int launch_maxpool2_stats(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s, double* partial, int G);
int launch_upsample2_stats(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, hipStream_t s, double* partial, int G);

// out = a * b * c * d (+ base)   elementwise over [M][C] with per-tensor strides; null pointers are skipped
// out = base + xt * (0.25 * sum of the four bilinearly upsampled S x S gate maps, S = 1,2,4,8, stored back to back) * col * edge
void launch_avif_combine(const float* base, const float* xt, const float* gates, const float* col, const float* edge,
                         float* out, int B, int H, int W, int C, hipStream_t s);
void launch_mul4_add(const float* base, int ldb, const float* a, int lda, const float* b, int ldb2, const float* c,
                     int ldc, const float* d, int ldd, float* out, int ldo, int64_t M, int C, hipStream_t s);

// time embedding (webp_inference.py:145-151) + all per-block time projections (webp_inference.py:308)
struct TimeEmbedW {
  const float* w0; const float* b0;   // [1024][256], [1024]
  const float* w2; const float* b2;   // [256][1024], [256]
  const float* wt; const float* bt;   // concatenated time_proj: [NT][256], [NT]
  int NT;
};
void launch_time_embed(const float* t, int B, const TimeEmbedW& w, float* hidden, float* temb, float* tproj,
                       hipStream_t s);

// boost[b] = clamp(a0 + a1*(1 - cl[b]), lo, hi)
void launch_cl_boost(const float* cl, int B, float a0, float a1, float lo, float hi, float* out, hipStream_t s);

// layout changes for the 3-channel image at the boundary
void launch_nchw3_to_nhwc4(const float* in, float* out, int B, int H, int W, hipStream_t s);
// [B*H*W][4] fp32 -> tanh -> [B][3][H][W]
void launch_nhwc4_to_nchw3_tanh(const float* in, float* out, int B, int H, int W, hipStream_t s);
// conv_tile.hip: out = conv3x3(act(x * scale[b][c] + shift[b][c])) + epilogue, the GroupNorm-apply pass fused into a full-resolution
// 3x3 conv with 64 output channels; p = the conv as launch_gemm16 would get it minus its A operand; x = the RAW tensor [B*H*W][ldx]
// (fp32, or bf16 when in16); raw16 != null: also writes a bf16 copy of x (the 1x1 shortcut's operand)
bool conv_tile_supported(const IGemm& p);
void launch_conv_tile(const IGemm& p, const void* x, int ldx, bool in16, const float* scale, const float* shift, int act, void* raw16,
                      int ldraw, hipStream_t s);
// head_fused.hip: tanh(conv3x3_{64->3}(SiLU(x * scale + shift)) + bias) in one pass over x [B*H*W][ldx] (fp32, or bf16 when in16);
// w16 = the packed head conv [9][4][64] bf16; out NCHW fp32.  H % 8 == 0 and W % 32 == 0.
bool head_fused_supported(int H, int W);
void launch_head_fused(const void* x, int ldx, bool in16, const float* scale, const float* shift, const void* w16, const float* bias,
                       float* out, int B, int H, int W, hipStream_t s);

// ------------------------------------------------------------------------------------------------ sampler
// u8 = uint8(trunc(clamp(x*127.5+127.5, 0, 255)))  (two roundings, no fma)   webp_inference.py:509
void launch_quantize_u8(const float* x, uint8_t* u8, int64_t n, hipStream_t s);
// DDRM update (webp_inference.py:584-600); dec_u8 is the decoded codec output.  noise may be null (i==0).
void launch_ddrm_update(const float* x_theta, const uint8_t* dec_u8, const float* y, const float* noise,
                        float* x_out, int64_t n_per_image, int B, float eta, float eta_b, float one_minus_eta_b,
                        float noise_coeff, int step_i, int steps, int last, hipStream_t s);
void launch_fill_t(float* t, int B, int step_i, int steps, hipStream_t s);
// Philox4x32-10 + Box-Muller standard normal, counter = global element index (independent of GPU count)
void launch_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s);

// Per-timestep scalars of one micro-batch lane, in DEVICE memory.  Every kernel of a timestep reads the loop index i and
// what derives from it from here, and the step's last kernel advances i, so a captured timestep (hipGraph) is replayed
// unchanged T times with no host-side parameter update (webp_inference.py:566-600 is the loop body).
struct StepParams {
  int i, steps, last;                  // loop index of the step about to run; last = (i == 0): x_t = x' without noise
  float t, noise_scale, coeff;         // t = float(i)/steps; noise_scale = t * coeff
  unsigned long long seed_base, seed;  // Philox key of step i = seed_base + i * golden-ratio constant
  unsigned long long rng_offset;       // Philox counter base of this lane: (global index of its first image) * 3HW / 4
  const float* noise; long long noise_stride;   // injected noise [steps][noise_stride] (null = Philox)
};
void launch_step_params_set(StepParams* p, int i, int steps, float coeff, uint64_t seed_base, uint64_t rng_offset,
                            const float* noise, int64_t noise_stride, hipStream_t s);
void launch_step_params_advance(StepParams* p, hipStream_t s);
void launch_fill_t_p(float* t, int B, const StepParams* p, hipStream_t s);
void launch_randn_p(float* out, int64_t n, const StepParams* p, hipStream_t s);
// dev_noise: Philox output for this lane (used when p->noise is null); lane_off: element offset of the lane inside one
// step of the injected noise tensor
void launch_ddrm_update_p(const float* x_theta, const uint8_t* dec_u8, const float* y, const float* dev_noise, int64_t lane_off,
                          float* x_out, int64_t n, float eta, float eta_b, float one_minus_eta_b, const StepParams* p,
                          hipStream_t s);

// on-device JPEG encode->decode round trip, bit-exact to libjpeg-turbo/Pillow defaults (jpeg_sim.hip)
size_t jpeg_sim_workspace(int B, int H, int W);
void launch_jpeg_roundtrip(const uint8_t* in, uint8_t* out, uint8_t* scratch, int B, int H, int W, int quality,
                           hipStream_t s, int force_sub = -1);   // force_sub = 1: 4:2:0 at every quality (Pillow's JPEG default)

// phase_consistency (webp_inference.py:531-550) on NCHW planes via separable DFTs.
struct DftPlan {           // device twiddle tables
  const float* cos_h; const float* sin_h; int H;
  const float* cos_w; const float* sin_w; int W;
};
// forward 2-D DFT of `planes` real images [P][H][W] -> re/im [P][H][W]; tmp_* are [P][H][W] scratch
void launch_dft2_forward(const float* x, float* re, float* im, float* tmp_re, float* tmp_im, int P,
                         const DftPlan& d, hipStream_t s);
// unit phase of fft2(y) computed in double (rows, then columns fused with the normalisation); tmp_*: [P][H][W] doubles
struct DftPlan64 { const double* cos_h; const double* sin_h; int H; const double* cos_w; const double* sin_w; int W; };
void launch_ref_unit_phase_f64(const float* y, float* ure, float* uim, double* tmp_re, double* tmp_im, int P, const DftPlan64& d,
                               hipStream_t s);
// unit phase of the reference spectrum: (re,im)/|.|, (1,0) where |.|==0
void launch_unit_phase(float* re, float* im, int64_t n, hipStream_t s);
// out = alpha*x + (1-alpha) * Re(ifft2(|X| * U))   (X = re/im in, U = unit phase)
void launch_phase_apply_inverse(const float* x, const float* xre, const float* xim, const float* ure,
                                const float* uim, float* tmp_re, float* tmp_im, float* tmp2_re, float* tmp2_im,
                                float* out, int P, const DftPlan& d, float alpha, float one_minus_alpha, hipStream_t s);

// eps-DDPM posterior mean step (conv.ipynb c1:124): x = c1 * (x - c2 * eps)
void launch_axpby(const float* x, const float* e, float* out, float c1, float c2, int64_t n, hipStream_t s);

}  // namespace ddpmir

// Host-side model of the restoration U-Nets: packed weights, workspace plan, forward, sampling loops.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ddpmir.h"
#include "host_codec.h"
#include "kernels.h"

// layout version of the packed weight blob (layout_weights): bump on any change of order, padding or derived tensors
#define DDPMIR_WEIGHT_LAYOUT_VERSION 3

namespace ddpmir {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

// bump allocator over one device allocation; dry run (base == nullptr) only measures
// A tensor of the activation stream BETWEEN residual blocks (block outputs, pooled / upsampled tensors, concat buffers): fp32,
// or bf16 where the bf16 activation stream is on (bf16 / fp8 mode of the DCT families; model.hip Context::stream16()).
struct ActBuf {
  void* p = nullptr; int ld = 0; bool h = false;      // h: elements are bf16
  ActBuf at(size_t elems) const { ActBuf a = *this; a.p = static_cast<char*>(p) + elems * (h ? 2 : 4); return a; }
  float* f() const { return static_cast<float*>(p); }
};
// Where norm1's statistics of a block input come from (all empty: a pass over the input).
struct InStats {
  const double* skip_sums = nullptr;   // up block: sums of the skip half, from the down block's conv_out epilogue
  int partial_splits = 0;              // bf16 stream: plan_.gn_partial holds the sums the max-pool (whole input) or the bilinear x2 (the
                                       // upsampled half of a concat input) left, written with this many splits per sample
  bool concat = false;                 // up block: input = cat[upsampled, skip]; its 8 groups split 4 + 4 over the halves
};

struct Arena {
  char* base = nullptr;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct ConvW {          // packed [taps][Cout][CinP] + bias
  float* w = nullptr; float* b = nullptr;
  void* w16 = nullptr;  // bf16 copy (DDPMIR_BF16 contexts)
  int cout = 0, cin = 0, cinp = 0, taps = 1;
  int cin16 = 0, cout16 = 0;   // strides of the bf16 copy: cinp/cout, or (Cin to 8, Cout to 4) for the DMA GEMM
  // DDPMIR_FP8 contexts, 3x3 convs with >= 256 input and output channels: e4m3 copy [tap][cout16][cin16] of w * 2^k
  // (k = the largest power of two that keeps max|w| * 2^k <= 448) and, in the blob, 2^-k for the epilogue
  void* w8 = nullptr; float* w8_inv_dev = nullptr; float w8_inv = 1.f;
  int w8_kexp = 0;          // host side, while packing: log2 of the weight scale
  bool is_f8 = false;       // this conv takes e4m3 operands (valid in the dry-run layout too, where w8 is still null)
};

struct AvifTransformW {
  float* tw = nullptr;  // [C][8][8]
  ConvW q0, q2;
};

struct BlockW {
  std::string name;
  int cin = 0, cinp = 0, cout = 0, level = 0;
  bool has_shortcut = false;
  float *n1g = nullptr, *n1b = nullptr, *n2g = nullptr, *n2b = nullptr;
  ConvW conv1, conv2, in_proj, out_proj, shortcut, conv_out;
  int tproj_off = 0;
  // WebP / JPEG frequency block
  float* dct = nullptr;
  ConvW low0, low2, high0, high2;
  float* co_bias = nullptr;   // conv_out.bias + shortcut.bias (shortcut folded into conv_out as a second K segment)
  void* sc16s = nullptr;      // DDPMIR_FP8, e4m3 conv_out: the shortcut's bf16 weights times conv_out's fp8 weight scale (a power of
                              // two), so the bf16 second K segment shares the accumulators and the one acc_scale of the fp8 taps
  ConvW fg1, fg2;   // DMA-GEMM path: both gates stacked (C -> [low hid | high hid]) and block-diagonal ([hid] -> [low | high])
  // AVIF frequency block
  AvifTransformW at;
  ConvW ms1[4], ms3[4], cc0, cc2, ep0, ep2;
};

// scalars baked into the launch arguments of a timestep (a captured timestep graph is valid for one such set)
struct LoopConsts {
  bool dev_codec = false;
  int quality = 0;
  float eta = 0.f, eta_b = 0.f, one_minus_eta_b = 0.f;
  double alpha = 0.0;
  bool operator==(const LoopConsts& o) const {
    return dev_codec == o.dev_codec && quality == o.quality && eta == o.eta && eta_b == o.eta_b &&
           one_minus_eta_b == o.one_minus_eta_b && alpha == o.alpha;
  }
};

struct GraphSlot {      // one captured launch sequence; `warm` = it has run eagerly once (lazy function attributes are set)
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool warm = false;
};

struct ArchConsts {
  int heads, dct_block, low_size;
  float boost_lo, boost_hi, head_mix, noise_coeff;
  int phase_q, phase_every;
  double phase_alpha;
};

class Context {
 public:
  explicit Context(const ddpmir_config& cfg);
  ~Context();

  std::string err;
  ddpmir_config cfg;
  ArchConsts k;

  void load_tensor(const std::string& name, const float* data, const int64_t* shape, int ndim);
  size_t weight_blob_bytes();
  void finalize_weights(void* dev_blob, int fill);
  int64_t param_count() const { return n_params_; }

  void predict(const float* x, const float* t, const float* cl, float* out, int B, int H, int W, hipStream_t s);
  void ddrm_sample_loop(const float* y, float* out, int B, int H, int W, int quality, int steps, double eta,
                        double eta_b, const ddpmir_loop_opts* opts);
  void phase_consistency(const float* x, const float* ref, float* out, int B, int H, int W, double alpha);
  void jpeg_roundtrip(const uint8_t* in, uint8_t* out, int B, int H, int W, int quality);
  void profile_enable(int enable);
  void profile_read(const std::string& cls, double* ms, int64_t* launches, double* flops);
  void set_option(const std::string& name, double value);

 private:
  // ---- weights
  std::map<std::string, HostTensor> host_;
  // The weight pointer table.  layout_weights() fills the table it is HANDED: the sizing pass (weight_blob_bytes) lays out a
  // scratch table over a base-less arena, finalize_weights lays out wt_ over the real blob -- so no dry run can ever re-point
  // the live table at NULL + offset (the round-3 memory-access fault).
  struct Weights {
    std::vector<BlockW> blocks;
    TimeEmbedW tw{};
    float *head_g = nullptr, *head_b = nullptr, *head_dct = nullptr;
    AvifTransformW head_at;
    ConvW head_conv;
    int64_t n_params = 0;
    int nt_total = 0;
  } wt_;
  std::vector<BlockW>& blocks_ = wt_.blocks;
  TimeEmbedW& tw_ = wt_.tw;
  float *&head_g_ = wt_.head_g, *&head_b_ = wt_.head_b, *&head_dct_ = wt_.head_dct;
  AvifTransformW& head_at_ = wt_.head_at;
  ConvW& head_conv_ = wt_.head_conv;
  int64_t& n_params_ = wt_.n_params;
  int& nt_total_ = wt_.nt_total;
  char* wblob_ = nullptr;
  bool wblob_owned_ = false;
  bool finalized_ = false;
  int64_t layout_params_ = 0;      // scratch counter of the layout pass in progress (put / put_conv), copied into its table at the end
  size_t blob_bytes_ = 0;          // packed blob size, measured once per configuration (weight_blob_bytes)
  void layout_weights(Arena& a, bool upload, Weights& w);
  void put(Arena& a, bool upload, float*& dst, const std::string& name, size_t expect);
  void put_conv(Arena& a, bool upload, ConvW& cw, const std::string& name, int cout, int cin, int ksz);
  void put_gate_pair(Arena& a, bool upload, BlockW& b, const std::string& fg);

  // ---- workspace (planned per (B,H,W))
  struct Plan {
    int B = 0, H = 0, W = 0;
    char* base = nullptr; size_t bytes = 0;
    float* x4 = nullptr;
    ActBuf cat[5];   // concat inputs of up5..up1, index = level
    ActBuf pin[6];   // pooled inputs, index = level (1..5)
    ActBuf bb0, bb1, bb2;
    ActBuf u[5];     // outputs of up blocks, index = level
    float *hA = nullptr, *hB = nullptr, *qkv = nullptr, *att = nullptr, *xd = nullptr, *hid = nullptr, *S = nullptr;
    float *xt = nullptr, *avg = nullptr, *col = nullptr, *edge = nullptr, *pool = nullptr, *pool8 = nullptr, *pg1 = nullptr, *pg2 = nullptr;
    unsigned char* attn_ws = nullptr;
    // bf16 operand tensors of the DMA-streamed GEMM path (dtype = bf16, DCT families)
    unsigned char *a8 = nullptr, *S8 = nullptr;      // e4m3 operand tensors of the fp8 convs (DDPMIR_FP8)
    unsigned short *a16 = nullptr, *x16 = nullptr, *hB16 = nullptr, *att16 = nullptr, *xdl16 = nullptr, *xdh16 = nullptr,
                   *hid16 = nullptr, *S16 = nullptr;
    unsigned char* zero16 = nullptr;
    float* splitk_ws = nullptr;
    double* gn_partial = nullptr;
    double* gn_sums = nullptr;   // [B][HW/32][8][2] statistics emitted by a GEMM epilogue (conv1 -> norm2)
    double* skip_sums[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [B][HW_l/32][4][2]: a down block's output -> the up block's norm1
    float *sc1 = nullptr, *sh1 = nullptr, *sc2 = nullptr, *sh2 = nullptr;
    float *hidden = nullptr, *temb = nullptr, *tproj = nullptr;
    float *boost = nullptr, *boost2 = nullptr, *constmix = nullptr;
    // sampler state
    float *s_xt = nullptr, *s_theta = nullptr, *s_y = nullptr, *s_noise = nullptr, *s_t = nullptr;
    StepParams* sp = nullptr;          // [2] per-lane step scalars on the device
    GraphSlot g_front[2], g_back[2][2];   // captured timestep halves per lane (back: without / with the phase fix)
    GraphSlot g_head[2], g_mid, g_tail[2];  // two-lane loops: down1 per lane, the levels below for both, up5 + head + D2H per lane
    LoopConsts lg_key; int lg_nl = 0;
    uint8_t *s_u8 = nullptr, *s_dec = nullptr, *s_jpeg = nullptr;
    float* dft[10] = {nullptr};
    float *tw_ch = nullptr, *tw_sh = nullptr, *tw_cw = nullptr, *tw_sw = nullptr;
    double *tw64_ch = nullptr, *tw64_sh = nullptr, *tw64_cw = nullptr, *tw64_sw = nullptr, *dft64_re = nullptr, *dft64_im = nullptr;
    // graph
    hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
    float *g_x = nullptr, *g_t = nullptr, *g_cl = nullptr, *g_out = nullptr;
  } plan_;
  void ensure_plan(int B, int H, int W);
  // fills the plan it is handed (ensure_plan sizes with a scratch Plan over a base-less arena, then lays out plan_ over the allocation)
  void layout_plan(Arena& a, Plan& p, int B, int H, int W) const;
  void free_plan();
  void order_after_caller();   // stream_ waits for what the caller enqueued on the default stream
  // ---- two concurrent lanes of the device-codec loop (model.hip: loop_two_lanes)
  std::unique_ptr<Context> lane_;           // the second lane: a context of its own (stream, workspace) on this context's weight blob
  bool is_lane_ = false;
  int lanes_opt_ = -1;                      // option "concurrent_lanes": -1 / 0 off (default), 1 free-running lanes, 2 attention launches take turns
  hipEvent_t attn_wait_ = nullptr, attn_done_ = nullptr;   // around the next full-resolution attention launch (set by the lane scheduler)
  hipEvent_t lane_done_[2] = {nullptr, nullptr};           // recorded after down1's / up5's attention of this lane
  bool use_two_lanes(int B, int H, int W, const ddpmir_loop_opts* opts, bool dev_codec) const;
  void ensure_lane();
  void loop_two_lanes(const float* y, float* out, int B, int H, int W, int quality, int steps, double eta, double eta_b,
                      const ddpmir_loop_opts* opts);

  void forward(const float* x, const float* t, const float* cl, float* out, hipStream_t s);
  // forward() in three pieces (model.hip): the full-resolution blocks per row range, the levels below for the whole batch
  void fwd_head(const float* x, const float* t, const float* cl, int row0, int nb, bool whole, hipStream_t s);
  void fwd_mid(int B, bool whole, hipStream_t s);
  void fwd_tail(float* out, int row0, int nb, bool whole, hipStream_t s);
  void prof_close_forward(hipStream_t s);
  // out_stats: a down block leaves the GroupNorm sums of its output there (conv_out's epilogue); in_skip_stats: an up block
  // takes the statistics of the skip half of its concat input from there instead of reading that half again
  void run_block(const BlockW& bw, ActBuf in, int H, int W, ActBuf dest, hipStream_t s, double* out_stats = nullptr, InStats st = InStats());
  void run_block16(const BlockW& bw, ActBuf in, int H, int W, ActBuf dest, hipStream_t s, double* out_stats, InStats st);
  // bf16 / fp8 mode, DCT families: the activation stream between blocks is bf16 (the reference's autocast rounds every conv
  // output to bf16 too); AVIF and fp32 mode keep fp32
  bool stream16() const;
  int stream16_opt_ = -1;   // option "bf16_activation_stream": -1 fp8 mode only (default), 0 never, 1 bf16 mode too
  int upsample(ActBuf in, int C, ActBuf cat, int B, int Hi, int Wi, hipStream_t s);
  bool skip_stats_fused(int level, int B, int H, int W) const;
  void conv16(const ConvW& cw, IGemm p, hipStream_t s, const float* bias_override = nullptr);
  bool use_gemm16() const { return cfg.dtype == DDPMIR_BF16 || cfg.dtype == DDPMIR_FP8; }
  void conv8(const ConvW& cw, IGemm p, hipStream_t s, const float* bias_override = nullptr);   // e4m3 operands (p.A16 = e4m3 tensor)
  int tile256_opt_ = -1;    // option "gemm256_tile" (IGemm::tile256)
  int epi_direct_opt_ = 1;  // option "epilogue_direct" (IGemm::epi_direct_opt)
  int head_fused_opt_ = 1;  // option "head_fused"
  int conv_tile_opt_ = 1;   // option "conv_tile"
  bool conv_tile_try(const ConvW& cw, IGemm p, const void* x, int ldx, bool in16, const float* scale, const float* shift, int act,
                     void* raw16, int ldraw, hipStream_t s);
  int fp8_min_ch_ = 256;    // DDPMIR_FP8: 3x3 convs with at least this many input AND output channels take e4m3 operands
  void freq_dct(const BlockW& bw, int H, int W, hipStream_t s);
  void freq_avif(const BlockW& bw, int H, int W, hipStream_t s);
  void avif_transform(const AvifTransformW& at, const float* x, int ldx, int C, int H, int W, float* T, float* tmp,
                      float* out, int ldo, const float* rowscale, const float* res, int ldres, hipStream_t s);
  void conv(const ConvW& cw, IGemm p, hipStream_t s);
  void phase_fix(float* x, int B, int H, int W, double alpha, hipStream_t s, size_t off);   // uses cached unit phase of y
  void phase_prepare(const float* y, int B, int H, int W, hipStream_t s, size_t off);
  void step_front(int l, int nb, const LoopConsts& c, hipStream_t s);
  void step_back(int l, int nb, bool with_phase, const LoopConsts& c, hipStream_t s);
  template <typename F> void run_or_replay(GraphSlot& g, bool use_graph, hipStream_t s, F&& body);
  void drop_loop_graphs();
  double codec_host_ms_ = 0.0; int64_t codec_calls_ = 0;   // host time spent inside the codec callback ("codec_host" class)
  int fb_ = 0;                      // batch of the forward being enqueued (<= plan_.B; micro-batch lanes)
  hipEvent_t lane_ev_[2] = {nullptr, nullptr};

  // ---- HIP-event profiling of kernel classes
  struct ProfSpan { hipEvent_t a, b; int cls; double flops; int count = 1; };
  std::vector<ProfSpan> prof_spans_;
  std::vector<hipEvent_t> prof_pool_;
  bool prof_on_ = false, prof_open_ = false;
  int prof_detail_ = 0;             // 1 = forward + full-resolution attention spans, 2 = every attention launch, 3 = + every GEMM launch
  static constexpr int kProfClasses = 5;
  double prof_ms_[kProfClasses] = {}; int64_t prof_n_[kProfClasses] = {}; double prof_flops_[kProfClasses] = {};
  // attention routing (attention_bf16.hip): score bound of the no-max kernel, and -- while profiling -- how many workgroups of
  // the general (running-max) kernel had work to do out of those launched beside a no-max kernel
  float attn_nm_bound_ = 80.0f;
  unsigned* attn_general_taken_ = nullptr;   // device counter
  int64_t attn_general_total_ = 0;
  AttnOpts attn_opts();
  hipEvent_t prof_event();
  void prof_begin(int cls, double flops, hipStream_t s);
  void prof_end(hipStream_t s);
  void prof_collect();

  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  uint8_t *pin_in_ = nullptr, *pin_out_ = nullptr; size_t pin_bytes_ = 0;
};

}  // namespace ddpmir

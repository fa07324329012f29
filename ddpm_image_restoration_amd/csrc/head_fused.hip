// Output head of the U-Net in one pass (bf16 / fp8 modes, DCT families):
//   out = tanh( conv3x3_{64 -> 3}( SiLU( GroupNorm(8)(x) ) ) + bias )          webp_inference.py:362-367, 399  (svd.ipynb c0:329-334)
// replacing three launches of the generic path -- the GroupNorm-apply pass that writes the bf16 operand (reads the mixed tensor,
// writes 2 bytes x 64 channels per pixel), the 3x3 conv through the strip kernel with its 3 output channels padded to a
// 32-column MFMA tile (which re-stages every input pixel three times, once per kernel row, plus the weights per tile), and the
// NHWC4 -> NCHW + tanh pass.  At 512 x 512, B = 8 those were 172 + 197 + 11 us (profiles/r04_c5_bf16_forward_timeline.txt); this
// kernel reads the mixed tensor ONCE.
//
// A workgroup (256 threads, 4 waves) owns an 8 x 32 pixel tile of one image.  Load phase: the (8 + 2) x (32 + 2) halo tile of
// 64 channels goes through registers -- scale / shift of the folded GroupNorm, SiLU, round to bf16: exactly the operand the
// generic path's gn_apply16 pass writes -- into LDS as [pixel][64 bf16] rows of 128 bytes; pixels outside the image are zeros
// (the conv's zero padding applies to the ACTIVATED tensor).  16-byte chunk c of pixel p sits at slot c ^ ((p >> 1) & 7), so both
// the stores (8 lanes per pixel) and the MFMA operand reads (a lane per pixel: 16 consecutive pixels = 16 distinct slots of the
// two 128-byte halves of a bank row) are conflict-free.  Compute phase: wave w owns tile rows 2w, 2w + 1; per row one
// v_mfma_f32_32x32x16_bf16 chain over K = 9 taps x 64 channels with A = the weights (32 "rows" of which 3 are real; the others
// read a zero row) and B = the row's 32 pixels shifted by the tap, so lane lr < 32 ends with its pixel's three outputs in
// acc[0..2].  Epilogue: + bias, tanh, three coalesced 128-byte row stores into the NCHW image.
#include <type_traits>

#include "kernels.h"

namespace ddpmir {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TH = 8, TW = 32;                      // output tile
constexpr int PW = TW + 2, PH = TH + 2;             // with halo
constexpr int ACT_BYTES = PH * PW * 128;            // 43 520
constexpr int WROWS = 5;                            // weight rows per tap in LDS: 4 (3 real + the padding row of the packed layout) + a zero row
constexpr int W_BYTES = 9 * WROWS * 128;            // 5 760

template <bool IN16>
__global__ __launch_bounds__(256) void head_fused_k(const void* __restrict__ xin, int ldx, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const __bf16* __restrict__ w16 /* [9][4][64] */,
                                                    const float* __restrict__ bias, float* __restrict__ out, int H, int W) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[ACT_BYTES + W_BYTES];
  unsigned char* wl = lds + ACT_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH, b = blockIdx.z;
  const size_t img = (size_t)b * H * W;
  // ---- weights: [tap][row 0..3][64 bf16] from the packed tensor, row 4 = zeros
  for (int i = tid; i < 9 * WROWS * 8; i += 256) {
    const int tap = i / (WROWS * 8), rem = i - tap * (WROWS * 8), row = rem >> 3, ch = rem & 7;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
    if (row < 4) v = *reinterpret_cast<const bf16x8*>(w16 + ((size_t)tap * 4 + row) * 64 + ch * 8);
    *reinterpret_cast<bf16x8*>(wl + (tap * WROWS + row) * 128 + ch * 16) = v;
  }
  // ---- activated halo tile.  A thread's pieces (8 channels of one pixel) are 256 apart: the same channel chunk for all of them, so
  // scale / shift are fetched once; all of a thread's loads are issued before the first is used (a loop that waits for each load
  // in turn is 11 dependent memory round trips per tile).
  {
    constexpr int NPIECE = PH * PW * 8, NI = (NPIECE + 255) / 256;
    const int ch = tid & 7, c = ch * 8;
    const float4 s0 = *reinterpret_cast<const float4*>(scale + (size_t)b * 64 + c), s1 = *reinterpret_cast<const float4*>(scale + (size_t)b * 64 + c + 4);
    const float4 f0 = *reinterpret_cast<const float4*>(shift + (size_t)b * 64 + c), f1 = *reinterpret_cast<const float4*>(shift + (size_t)b * 64 + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    typename std::conditional<IN16, bf16x8, f32x8>::type raw[NI];
    unsigned inside = 0;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256, p = i >> 3;
      const int py = p / PW, px = p - py * PW;
      const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
      if (i < NPIECE && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        inside |= 1u << it;
        const size_t m = img + (size_t)gy * W + gx;
        if constexpr (IN16) raw[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(xin) + m * ldx + c);
        else raw[it] = *reinterpret_cast<const f32x8*>(reinterpret_cast<const float*>(xin) + m * ldx + c);
      }
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256, p = i >> 3;
      if (i >= NPIECE) continue;
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.f;
      if (inside & (1u << it)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)act16(fmaf((float)raw[it][j], sc[j], sh[j]), ACT_SILU);   // = gn_apply16_k's operand
      }
      *reinterpret_cast<bf16x8*>(lds + p * 128 + ((ch ^ ((p >> 1) & 7)) << 4)) = o;
    }
  }
  __syncthreads();
  // ---- 2 tile rows per wave, one MFMA chain each
  const int lr = lane & 31, lh = lane >> 5;
  const int wrow = lr < 4 ? lr : 4;                    // weight row of this lane's A fragment (rows >= 4: zeros)
  f32x16 acc[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap - dy * 3;         // (dy + 1, dx + 1) - 1: offsets into the halo tile
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(wl + (tap * WROWS + wrow) * 128 + (2 * ks + lh) * 16);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int p = (2 * wave + g + dy) * PW + lr + dx;
        const bf16x8 bq = *reinterpret_cast<const bf16x8*>(lds + p * 128 + (((2 * ks + lh) ^ ((p >> 1) & 7)) << 4));
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[g], 0, 0, 0);
      }
    }
  }
  // acc[g][r], lane (lr, lh): output channel (r & 3) + 8 (r >> 2) + 4 lh of pixel lr  ->  lanes 0..31 hold channels 0..2 in acc[g][0..2]
  if (lh == 0) {
    const int gx = tx0 + lr;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int gy = ty0 + 2 * wave + g;
#pragma unroll
      for (int o = 0; o < 3; ++o)
        out[(((size_t)b * 3 + o) * H + gy) * W + gx] = tanhf(acc[g][o] + bias[o]);
    }
  }
}

}  // namespace

bool head_fused_supported(int H, int W) { return H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0; }

// x: the mixed tensor [B*H*W][ldx] (fp32, or bf16 when in16); scale / shift [B][64]; w16 [9][4][64] bf16 (the packed head conv:
// Cout padded to 4, Cin 64); bias [>= 3]; out NCHW fp32 [B][3][H][W]
void launch_head_fused(const void* x, int ldx, bool in16, const float* scale, const float* shift, const void* w16, const float* bias,
                       float* out, int B, int H, int W, hipStream_t s) {
  if (!head_fused_supported(H, W)) throw std::runtime_error("head_fused: image sides must be multiples of 8 x 32");
  const dim3 grid(W / TW, H / TH, B);
  if (in16) hipLaunchKernelGGL((head_fused_k<true>), grid, dim3(256), 0, s, x, ldx, scale, shift, reinterpret_cast<const __bf16*>(w16), bias, out, H, W);
  else hipLaunchKernelGGL((head_fused_k<false>), grid, dim3(256), 0, s, x, ldx, scale, shift, reinterpret_cast<const __bf16*>(w16), bias, out, H, W);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

// GroupNorm-apply (+ activation) FUSED INTO the 3x3 convolution that consumes it, for the full-resolution blocks (64 output
// channels, 64 k input channels):        out = conv3x3( act( x * scale[b][c] + shift[b][c] ) ) + bias (+ time bias)
//   norm1 -> conv1  webp_inference.py:306-309      norm2 -> GELU -> conv2  webp_inference.py:311-313
//
// The DMA-fed GEMM kernels (gemm16.hip) need their A operand as a finished bf16 tensor, so every GroupNorm costs a pass of its
// own (gn_apply16_k: read the raw tensor, write the normalised / activated bf16 operand) before the conv re-reads that operand
// three times (once per kernel row).  At 512 x 512, B = 8 that pair is 156 + 250 us for norm2 + conv2 of a 64-channel block and
// 436 + 449 us for norm1 + conv1 of up5 (profiles/r04_c5_bf16_forward_timeline.txt).  This kernel reads the RAW tensor once:
//
// * PERSISTENT, WEIGHTS-STATIONARY, WAVE-SPECIALISED.  One workgroup of 768 threads per CU keeps the whole weight tensor
//   (9 taps x 64 x 64 bf16 = 72 KiB, fetched once by LDS-DMA) in LDS and walks 8 x 32 pixel tiles.  Waves 4..11 are LOADERS (the transform is 30 VALU instructions per element: the GELU alone is 150 us of the whole chip's vector issue at 512 x 512, B = 8, so it gets eight of the twelve waves; sixteen waves would cap the registers at 128 and spill): the
//   (8 + 2) x (32 + 2) halo tile of the NEXT tile goes through their registers -- scale / shift, activation (the bf16-grade GELU
//   of gn_apply16_k: the SAME operand values), round to bf16 -- into one of two 42.5-KiB LDS buffers as [pixel][64 bf16] rows of
//   128 bytes; pixels outside the image are zeros (the conv pads the ACTIVATED tensor).  Waves 0..3 COMPUTE the current tile
//   from the other buffer: wave w owns tile rows 2w, 2w + 1; per tap 16 MFMAs (2 rows x 2 column tiles x 4 K steps), 144 per
//   tile.  One s_barrier per tile hands the buffers over, so a tile costs max(load, compute) instead of their sum, and the
//   loaders' GELU arithmetic runs on the SIMDs' vector pipes while the compute waves keep the matrix pipes busy.  (The first cut
//   of this kernel streamed the weights per tap through a double buffer, two workgroups per CU: nine exposed DMA round trips per
//   tile, 400 us per launch -- no faster than the two launches it replaced.)
// * Every input element is transformed 1.33 times (the halo) instead of being staged nine times.  16-byte chunk c of pixel p
//   sits at slot c ^ ((p >> 1) & 7): stores (8 lanes per pixel) and MFMA operand reads (a lane per pixel) are conflict-free
//   (same layout as head_fused.hip); weight rows are swizzled on the source side as in gemm16.hip / gemm256.hip.
// * A tile row's 32 pixels are the MFMA's 32 "rows", so the accumulators have the layout of the GEMM kernels and the
//   register-direct epilogue of gemm_epilogue.h (bias, time bias, fp32 / bf16 copies, fused GroupNorm partials of the OUTPUT:
//   one writer per 32-pixel row and group) is reused unchanged -- a tile row of 32 pixels starting at a multiple of 32 is
//   exactly one 32-row wave tile of the [B*H*W][C] matrix.
// * Optionally the raw input also leaves as a bf16 copy (interior pixels): the operand of a block's 1x1 shortcut.
//
// LDS: 72 KiB weights + 2 x 42.5 KiB tiles = 157 KiB of the CU's 160.
#include <algorithm>
#include <type_traits>

#include "kernels.h"
#include "gemm_epilogue.h"

namespace ddpmir {
namespace {

constexpr int TH = 8, TW = 32, PW = TW + 2, PH = TH + 2;
constexpr int ACT_BYTES = PH * PW * 128;            // 43 520: one activated halo tile (64 channels)
constexpr int WT_BYTES = 9 * 64 * 128;              // 73 728: all nine taps' weights, [tap][64 output channels][64 input channels]
constexpr int LDS_BYTES = WT_BYTES + 2 * ACT_BYTES; // 160 768

template <bool IN16>
__global__ __launch_bounds__(768, 1) void conv_tile_kernel(const IGemm p, const void* __restrict__ xin, int ldx, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, __bf16* __restrict__ raw16, int ldraw,
                                                           int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* wl = lds;                             // weights
  unsigned char* al = lds + WT_BYTES;                  // two activation tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const bool loader = wave_s >= 4;
  const int lr = lane & 31, lh = lane >> 5;
  const int H = p.H, W = p.W, HW = H * W, M = p.B * HW;
  const __bf16* w16 = reinterpret_cast<const __bf16*>(p.Wt16);

  // ---- weights, once: slot s = j * 512 + tid -> tap s / 512 ... (row n = output channel, the row's 16-byte chunk (s & 7) ^ ((n >> 1) & 7)
  // at linear position s & 7: source-side swizzle, the fragment read undoes it)
  {
    const unsigned wl0 = (unsigned)(size_t)wl;
#pragma unroll
    for (int j = 0; j < 6; ++j) {                      // 4608 slots = 9 taps x 64 rows x 8 chunks = 6 rounds of 768
      const int sidx = j * 768 + tid, tap = sidx >> 9, row = (sidx >> 3) & 63, chunk = (sidx & 7) ^ ((row >> 1) & 7);
      const __bf16* g = w16 + ((size_t)tap * 64 + row) * 64 + chunk * 8;
      __builtin_amdgcn_global_load_lds((const void*)g, (lds_ptr_t)(size_t)(wl0 + (j * 768 + wave_s * 64) * 16), 16, 0, 0);
    }
  }

  // tile index -> (image, tile row, tile column); consecutive tiles are neighbours in an image row (shared halo columns in L2)
  auto tile_pos = [&](int t, int& b, int& ty0, int& tx0) {
    const int per = tiles_x * tiles_y;
    b = t / per;
    const int r = t - b * per, ty = r / tiles_x;
    ty0 = ty * TH; tx0 = (r - ty * tiles_x) * TW;
  };
  // loader waves: activated halo tile of tile t into buffer buf.  A thread's pieces (16 bytes = 8 channels of one pixel) are 256
  // apart, so its channel chunk is the same for all of them: scale / shift are fetched once per tile; ALL of a thread's loads are
  // issued before the first is used (the first version of this loop waited for every load in turn: 11 dependent round trips,
  // 16 us per tile).
  constexpr int NLOAD = 512;                           // loader threads: waves 4 .. 11
  constexpr int NPIECE = PH * PW * 8, NI = (NPIECE + NLOAD - 1) / NLOAD;
  auto fill = [&](int t, int buf) {
    int b, ty0, tx0;
    tile_pos(t, b, ty0, tx0);
    const size_t img = (size_t)b * HW;
    unsigned char* dst = al + buf * ACT_BYTES;
    const int lt = tid - 256, ch = lt & 7, c = ch * 8;   // (NLOAD % 8 == 0: a thread's pieces all have the same channel chunk)
    float sc[8], sh[8];
    if (scale) {
      const float4 s0 = *reinterpret_cast<const float4*>(scale + (size_t)b * 64 + c), s1 = *reinterpret_cast<const float4*>(scale + (size_t)b * 64 + c + 4);
      const float4 f0 = *reinterpret_cast<const float4*>(shift + (size_t)b * 64 + c), f1 = *reinterpret_cast<const float4*>(shift + (size_t)b * 64 + c + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = f0.x; sh[1] = f0.y; sh[2] = f0.z; sh[3] = f0.w; sh[4] = f1.x; sh[5] = f1.y; sh[6] = f1.z; sh[7] = f1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
    }
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    constexpr int NB = NI;                             // all of a thread's loads in flight at once (6 pieces)
#pragma unroll
    for (int base = 0; base < NI; base += NB) {
      typename std::conditional<IN16, bf16x8, f32x8>::type raw[NB];
      unsigned flags = 0;                              // bit k: piece inside the image; bit 16 + k: interior pixel (raw copy)
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int i = lt + (base + k) * NLOAD;
        const int q = i >> 3, py = q / PW, px = q - py * PW;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
        const bool in = base + k < NI && i < NPIECE && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        if (in) {
          const size_t m = img + (size_t)gy * W + gx;
          flags |= 1u << k;
          if (py >= 1 && py <= TH && px >= 1 && px <= TW) flags |= 1u << (16 + k);
          if constexpr (IN16) raw[k] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(xin) + m * ldx + c);
          else raw[k] = *reinterpret_cast<const f32x8*>(reinterpret_cast<const float*>(xin) + m * ldx + c);
        }
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int i = lt + (base + k) * NLOAD;
        if (base + k >= NI || i >= NPIECE) continue;
        const int q = i >> 3;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.f;
        if (flags & (1u << k)) {
          float r[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = (float)raw[k][j];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (__bf16)act16(fmaf(r[j], sc[j], sh[j]), act);   // = gn_apply16_k's operand
          if (raw16 && (flags & (1u << (16 + k)))) {     // bf16 copy of the raw input (interior pixels): a shortcut's operand
            bf16x8 rq;
#pragma unroll
            for (int j = 0; j < 8; ++j) rq[j] = (__bf16)r[j];
            const int py = q / PW, px = q - py * PW;
            *reinterpret_cast<bf16x8*>(raw16 + (img + (size_t)(ty0 - 1 + py) * W + (tx0 - 1 + px)) * ldraw + c) = rq;
          }
        }
        *reinterpret_cast<bf16x8*>(dst + q * 128 + ((ch ^ ((q >> 1) & 7)) << 4)) = o;
      }
    }
  };
  // compute waves: tile t from buffer buf
  const int bkey = (lr >> 1) & 7;                      // swizzle key of this lane's weight row (nt * 32 + lr: the key only depends on lr)
  auto compute = [&](int t, int buf) {
    int b, ty0, tx0;
    tile_pos(t, b, ty0, tx0);
    const unsigned char* src = al + buf * ACT_BYTES;
    f32x16 acc[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][nt][r] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {                // (not unrolled: the nine taps' fragment reads hoisted together spill)
      const unsigned char* wb = wl + tap * (64 * 128);
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 bfr[2], af[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          bfr[nt] = *reinterpret_cast<const bf16x8*>(wb + (nt * 32 + lr) * 128 + (((2 * ks + lh) ^ bkey) << 4));
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int q = (2 * wave + g + dy) * PW + lr + dx;
          af[g] = *reinterpret_cast<const bf16x8*>(src + q * 128 + (((2 * ks + lh) ^ ((q >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g], bfr[nt], acc[g][nt], 0, 0, 0);
      }
    }
    // epilogue: a tile row = 32 consecutive pixels = one 32-row wave tile of the [B*H*W][Cout] matrix
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int mw0 = (int)((size_t)b * HW + (size_t)(ty0 + 2 * wave + g) * W + tx0);
      gemm16_epilogue_direct_any<2>(p, acc[g], mw0, 0, M, HW);
    }
  };

  // ---- the walk.  Every wave executes the same barriers: one after the first fill, one per tile.
  int t = blockIdx.x;
  if (loader && t < ntiles) fill(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weights (every wave issued a share) and the first tile have landed
  __syncthreads();
  for (int i = 0; t < ntiles; t += gridDim.x, ++i) {
    const int tn = t + (int)gridDim.x;
    if (loader) { if (tn < ntiles) fill(tn, (i + 1) & 1); }
    else compute(t, i & 1);
    __syncthreads();                                   // tile i + 1 is complete, tile i's buffer is free
  }
}

}  // namespace

// p: the conv as launch_gemm16 would get it (Wt16 [9][64][64] bf16, bias, tb, out / out16, gn_sums ...) WITHOUT its A operand
bool conv_tile_supported(const IGemm& p) {
  return p.taps == 9 && p.Cout == 64 && p.Cin == 64 && !p.f8 && !p.Cinb && (p.H % TH) == 0 && (p.W % TW) == 0 && p.Wt16 &&
         (int64_t)p.B * p.H * p.W < (int64_t(1) << 31) && epilogue_direct_mode(p, 64) != 0;
}

void launch_conv_tile(const IGemm& p, const void* x, int ldx, bool in16, const float* scale, const float* shift, int act, void* raw16,
                      int ldraw, hipStream_t s) {
  if (!conv_tile_supported(p)) throw std::runtime_error("conv_tile: unsupported shape or epilogue");
  static int n_cu = 0;
  static PerDeviceOnce once;
  once.run([] {
    int dev = 0;
    hipDeviceProp_t prop;
    DDPMIR_HIP_CHECK(hipGetDevice(&dev));
    DDPMIR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount;
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  });
  IGemm q = p;
  q.fd_hw = FastDiv((unsigned)(p.H * p.W));
  q.fd_w = FastDiv((unsigned)p.W);
  q.epi_direct = epilogue_direct_mode(p, 64);
  const int tx = p.W / TW, ty = p.H / TH, ntiles = tx * ty * p.B;
  const int grid = std::min(ntiles, n_cu > 0 ? n_cu : 256);       // one persistent workgroup per CU (157 KiB of LDS each)
  if (in16) hipLaunchKernelGGL((conv_tile_kernel<true>), dim3(grid), dim3(768), LDS_BYTES, s, q, x, ldx, scale, shift, act, reinterpret_cast<__bf16*>(raw16), ldraw, tx, ty, ntiles);
  else hipLaunchKernelGGL((conv_tile_kernel<false>), dim3(grid), dim3(768), LDS_BYTES, s, q, x, ldx, scale, shift, act, reinterpret_cast<__bf16*>(raw16), ldraw, tx, ty, ntiles);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

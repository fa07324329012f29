// On-device JPEG encode->decode round trip, bit-exact to libjpeg-turbo's default integer path as driven by
// Pillow (reference: jpeg_compress, svd.ipynb c0:45-69: quality clamp, 4:4:4 above q30 else 4:2:0).
//
// Entropy coding is lossless, so decode(encode(x)) only depends on: RGB->YCbCr (jccolor.c fixed point),
// h2v2 downsampling with alternating bias (jcsample.c), forward DCT jpeg_fdct_islow (jfdctint.c),
// quantisation by the quality-scaled Annex-K tables (jcparam.c, jcdctmgr.c), dequantisation + inverse DCT
// jpeg_idct_islow with the post-IDCT range limit (jidctint.c, jdmaster.c), h2v2 "fancy" triangle upsampling
// (jdsample.c) and YCbCr->RGB (jdcolor.c).  All of it is integer arithmetic, restated here from the published
// algorithms; tests compare it bit for bit with Pillow on random and synthetic images at several qualities.
// Image sizes are multiples of 32, so MCUs (16x16 for 4:2:0) tile the image exactly and no edge padding occurs.
#include "kernels.h"

namespace ddpmir {

namespace {

__constant__ unsigned char kStdLuma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                           14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                           18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                           49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
__constant__ unsigned char kStdChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                             24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270,
              F_0_899976223 = 7373, F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137,
              F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline = TRUE)
__device__ __forceinline__ int quant_value(int base, int quality) {
  int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
  int scale = q < 50 ? 5000 / q : 200 - 2 * q;
  int v = (base * scale + 50) / 100;
  return v < 1 ? 1 : (v > 255 ? 255 : v);
}

// post-IDCT range limit table of jdmaster.c (index masked to 10 bits)
__device__ __forceinline__ int idct_range_limit(int x) {
  int j = x & 1023;
  return j < 128 ? 128 + j : (j < 512 ? 255 : (j < 896 ? 0 : j - 896));
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one 1-D pass of jpeg_fdct_islow on 8 values (in place); pass 2 differs only in the descale amounts
__device__ __forceinline__ void fdct_1d(int* d, bool pass1) {
  int tmp0 = d[0] + d[7], tmp7 = d[0] - d[7], tmp1 = d[1] + d[6], tmp6 = d[1] - d[6];
  int tmp2 = d[2] + d[5], tmp5 = d[2] - d[5], tmp3 = d[3] + d[4], tmp4 = d[3] - d[4];
  int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  const int sh = pass1 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
  if (pass1) { d[0] = (tmp10 + tmp11) << PASS1_BITS; d[4] = (tmp10 - tmp11) << PASS1_BITS; }
  else { d[0] = descale(tmp10 + tmp11, PASS1_BITS); d[4] = descale(tmp10 - tmp11, PASS1_BITS); }
  int z1 = (tmp12 + tmp13) * F_0_541196100;
  d[2] = descale(z1 + tmp13 * F_0_765366865, sh);
  d[6] = descale(z1 + tmp12 * (-F_1_847759065), sh);
  z1 = tmp4 + tmp7;
  int z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
  int z5 = (z3 + z4) * F_1_175875602;
  tmp4 *= F_0_298631336; tmp5 *= F_2_053119869; tmp6 *= F_3_072711026; tmp7 *= F_1_501321110;
  z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
  z3 += z5; z4 += z5;
  d[7] = descale(tmp4 + z1 + z3, sh);
  d[5] = descale(tmp5 + z2 + z4, sh);
  d[3] = descale(tmp6 + z2 + z3, sh);
  d[1] = descale(tmp7 + z1 + z4, sh);
}

// one 1-D pass of jpeg_idct_islow; pass 1 (columns) keeps PASS1_BITS extra bits, pass 2 (rows) range-limits
__device__ __forceinline__ void idct_1d(int* d, bool pass1) {
  int z2 = d[2], z3 = d[6];
  int z1 = (z2 + z3) * F_0_541196100;
  int tmp2 = z1 + z3 * (-F_1_847759065);
  int tmp3 = z1 + z2 * F_0_765366865;
  z2 = d[0]; z3 = d[4];
  int tmp0 = (z2 + z3) << CONST_BITS, tmp1 = (z2 - z3) << CONST_BITS;
  int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = d[7]; tmp1 = d[5]; tmp2 = d[3]; tmp3 = d[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  int z5 = (z3 + z4) * F_1_175875602;
  tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
  z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  if (pass1) {
    const int sh = CONST_BITS - PASS1_BITS;
    d[0] = descale(tmp10 + tmp3, sh); d[7] = descale(tmp10 - tmp3, sh);
    d[1] = descale(tmp11 + tmp2, sh); d[6] = descale(tmp11 - tmp2, sh);
    d[2] = descale(tmp12 + tmp1, sh); d[5] = descale(tmp12 - tmp1, sh);
    d[3] = descale(tmp13 + tmp0, sh); d[4] = descale(tmp13 - tmp0, sh);
  } else {
    const int sh = CONST_BITS + PASS1_BITS + 3;
    d[0] = idct_range_limit(descale(tmp10 + tmp3, sh)); d[7] = idct_range_limit(descale(tmp10 - tmp3, sh));
    d[1] = idct_range_limit(descale(tmp11 + tmp2, sh)); d[6] = idct_range_limit(descale(tmp11 - tmp2, sh));
    d[2] = idct_range_limit(descale(tmp12 + tmp1, sh)); d[5] = idct_range_limit(descale(tmp12 - tmp1, sh));
    d[3] = idct_range_limit(descale(tmp13 + tmp0, sh)); d[4] = idct_range_limit(descale(tmp13 - tmp0, sh));
  }
}

// ---- stage 1: RGB (uint8 NCHW) -> Y plane [B][H][W] and Cb/Cr planes [B][Hc][Wc] (Hc = H/2 if sub)
__global__ void jpeg_rgb_to_ycc_k(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ yp, uint8_t* __restrict__ cb,
                                  uint8_t* __restrict__ cr, int B, int H, int W, int sub) {
  const int Hc = sub ? H / 2 : H, Wc = sub ? W / 2 : W;
  const int64_t total = (int64_t)B * Hc * Wc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int xc = i % Wc;
    int64_t t = i / Wc;
    int yc = t % Hc;
    int b = t / Hc;
    const uint8_t* R = rgb + ((size_t)b * 3 + 0) * H * W;
    const uint8_t* G = rgb + ((size_t)b * 3 + 1) * H * W;
    const uint8_t* Bl = rgb + ((size_t)b * 3 + 2) * H * W;
    const int n = sub ? 2 : 1;
    int scb = 0, scr = 0;
    for (int dy = 0; dy < n; ++dy)
      for (int dx = 0; dx < n; ++dx) {
        int y = yc * n + dy, x = xc * n + dx;
        int r = R[(size_t)y * W + x], g = G[(size_t)y * W + x], bl = Bl[(size_t)y * W + x];
        int Y = (19595 * r + 38470 * g + 7471 * bl + 32768) >> 16;
        int Cb = (-11059 * r - 21709 * g + 32768 * bl + (128 << 16) + 32767) >> 16;
        int Cr = (32768 * r - 27439 * g - 5329 * bl + (128 << 16) + 32767) >> 16;
        yp[((size_t)b * H + y) * W + x] = (uint8_t)Y;
        scb += Cb; scr += Cr;
      }
    if (sub) {   // h2v2_downsample: bias alternates 1,2,1,2 along the output row
      int bias = (xc & 1) ? 2 : 1;
      scb = (scb + bias) >> 2; scr = (scr + bias) >> 2;
    }
    cb[((size_t)b * Hc + yc) * Wc + xc] = (uint8_t)scb;
    cr[((size_t)b * Hc + yc) * Wc + xc] = (uint8_t)scr;
  }
}

// ---- stage 2: per 8x8 block fdct -> quantise -> dequantise -> idct, in place on a uint8 plane
__global__ void jpeg_block_roundtrip_k(uint8_t* __restrict__ plane, int P, int H, int W, int chroma, int quality) {
  const int nbx = W / 8, nby = H / 8;
  const int64_t total = (int64_t)P * nby * nbx;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int bx = i % nbx;
    int64_t t = i / nbx;
    int by = t % nby;
    int p = t / nby;
    uint8_t* base = plane + ((size_t)p * H + by * 8) * W + bx * 8;
    int d[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)r * W);
#pragma unroll
      for (int c = 0; c < 4; ++c) { d[r][c] = (int)((v.x >> (8 * c)) & 255u) - 128; d[r][4 + c] = (int)((v.y >> (8 * c)) & 255u) - 128; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) fdct_1d(d[r], true);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int col[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) col[r] = d[r][c];
      fdct_1d(col, false);
#pragma unroll
      for (int r = 0; r < 8; ++r) d[r][c] = col[r];
    }
    // quantise (round half away from zero on the x8-scaled DCT output) and dequantise
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        int qv = quant_value(chroma ? kStdChroma[r * 8 + c] : kStdLuma[r * 8 + c], quality);
        int div = qv << 3;
        int v = d[r][c];
        int a = v < 0 ? -v : v;
        a = (a + (div >> 1)) / div;
        d[r][c] = (v < 0 ? -a : a) * qv;
      }
    // inverse: columns first (pass 1), then rows (pass 2)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int col[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) col[r] = d[r][c];
      idct_1d(col, true);
#pragma unroll
      for (int r = 0; r < 8; ++r) d[r][c] = col[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      idct_1d(d[r], false);
      uint2 v;
      v.x = (unsigned)d[r][0] | ((unsigned)d[r][1] << 8) | ((unsigned)d[r][2] << 16) | ((unsigned)d[r][3] << 24);
      v.y = (unsigned)d[r][4] | ((unsigned)d[r][5] << 8) | ((unsigned)d[r][6] << 16) | ((unsigned)d[r][7] << 24);
      *reinterpret_cast<uint2*>(base + (size_t)r * W) = v;
    }
  }
}

// h2v2 fancy (triangle) upsampling of one chroma sample position, jdsample.c h2v2_fancy_upsample
__device__ __forceinline__ int fancy_up(const uint8_t* __restrict__ c, int Hc, int Wc, int y, int x) {
  const int yc = y >> 1, xc = x >> 1;
  const int yn = (y & 1) ? min(yc + 1, Hc - 1) : max(yc - 1, 0);   // nearer row yc (3/4), further row yn (1/4)
  auto colsum = [&](int xx) { return (int)c[(size_t)yc * Wc + xx] * 3 + (int)c[(size_t)yn * Wc + xx]; };
  const int cur = colsum(xc);
  if ((x & 1) == 0) {
    if (xc == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + colsum(xc - 1) + 8) >> 4;
  }
  if (xc == Wc - 1) return (cur * 4 + 7) >> 4;
  return (cur * 3 + colsum(xc + 1) + 7) >> 4;
}

// ---- stage 3: (upsample) + YCbCr -> RGB uint8 NCHW, jdcolor.c ycc_rgb_convert
__global__ void jpeg_ycc_to_rgb_k(const uint8_t* __restrict__ yp, const uint8_t* __restrict__ cb,
                                  const uint8_t* __restrict__ cr, uint8_t* __restrict__ rgb, int B, int H, int W,
                                  int sub) {
  const int Hc = sub ? H / 2 : H, Wc = sub ? W / 2 : W;
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int x = i % W;
    int64_t t = i / W;
    int y = t % H;
    int b = t / H;
    int Y = yp[i];
    const uint8_t* cbp = cb + (size_t)b * Hc * Wc;
    const uint8_t* crp = cr + (size_t)b * Hc * Wc;
    int Cb, Cr;
    if (sub) { Cb = fancy_up(cbp, Hc, Wc, y, x); Cr = fancy_up(crp, Hc, Wc, y, x); }
    else { Cb = cbp[(size_t)y * W + x]; Cr = crp[(size_t)y * W + x]; }
    const int cbx = Cb - 128, crx = Cr - 128;
    int r = Y + ((91881 * crx + 32768) >> 16);
    int g = Y + ((-22554 * cbx + 32768 - 46802 * crx) >> 16);
    int bl = Y + ((116130 * cbx + 32768) >> 16);
    rgb[((size_t)b * 3 + 0) * H * W + (size_t)y * W + x] = (uint8_t)clamp255(r);
    rgb[((size_t)b * 3 + 1) * H * W + (size_t)y * W + x] = (uint8_t)clamp255(g);
    rgb[((size_t)b * 3 + 2) * H * W + (size_t)y * W + x] = (uint8_t)clamp255(bl);
  }
}

inline unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (unsigned)g;
}

}  // namespace

size_t jpeg_sim_workspace(int B, int H, int W) { return (size_t)B * H * W * 3 + 1024; }

// in/out: uint8 NCHW device pointers; scratch: jpeg_sim_workspace() bytes
// force_sub: -1 = the JPEG family's rule (4:4:4 above q30, svd.ipynb c0:59); 1 = 4:2:0 whatever the quality -- Pillow's default
// for a plain save(format="JPEG", quality=q), which is what the AVIF family's per-image fallback does (avif_inference.py:88-95)
void launch_jpeg_roundtrip(const uint8_t* in, uint8_t* out, uint8_t* scratch, int B, int H, int W, int quality,
                           hipStream_t s, int force_sub) {
  if ((H % 16) != 0 || (W % 16) != 0) throw std::runtime_error("jpeg simulator: H and W must be multiples of 16");
  const int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);        // svd.ipynb c0:57
  const int sub = force_sub >= 0 ? force_sub : (q > 30 ? 0 : 1);         // svd.ipynb c0:59
  const int Hc = sub ? H / 2 : H, Wc = sub ? W / 2 : W;
  uint8_t* yp = scratch;
  uint8_t* cb = yp + (size_t)B * H * W;
  uint8_t* cr = cb + (size_t)B * Hc * Wc;
  hipLaunchKernelGGL(jpeg_rgb_to_ycc_k, dim3(grid_for((int64_t)B * Hc * Wc, 256)), dim3(256), 0, s, in, yp, cb, cr, B, H,
                     W, sub);
  hipLaunchKernelGGL(jpeg_block_roundtrip_k, dim3(grid_for((int64_t)B * (H / 8) * (W / 8), 64)), dim3(64), 0, s, yp, B, H,
                     W, 0, q);
  hipLaunchKernelGGL(jpeg_block_roundtrip_k, dim3(grid_for((int64_t)2 * B * (Hc / 8) * (Wc / 8), 64)), dim3(64), 0, s, cb,
                     2 * B, Hc, Wc, 1, q);
  hipLaunchKernelGGL(jpeg_ycc_to_rgb_k, dim3(grid_for((int64_t)B * H * W, 256)), dim3(256), 0, s, yp, cb, cr, out, B, H, W,
                     sub);
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

// 256 x 256 x (128 bytes of K) implicit-GEMM tile with EIGHT waves for the 3x3 convolutions with Cout >= 256
// (webp_inference.py:282, 292, 229 at levels 2-5), bf16 and OCP e4m3 operands, gfx950.
//
//   out[m][n] = epi( sum_{c-chunk, tap} A[pixel(m) + tap][c] * W[tap][n][c]  (+ sum_c A2[m][c] * W2[n][c]) )
//
// gemm16_kernel (128-row tiles, four waves, two-stage ring, three or four workgroups per CU) runs these shapes at 0.63-0.69
// PFLOP/s in bf16 and ~0.93 in e4m3: half of every wave's cycles are spent parked at the stage wait (profiles/r03_gemm_pmc.txt).
// This kernel is the structure that overlaps the stage waits with MFMAs INSIDE one workgroup (the guide's 8-phase template;
// measured stand-alone in round 3, tools/micro/gemm256_stagger.hip: 0.88-0.94 PFLOP/s bf16, 1.7-1.9 e4m3 on the same shapes):
//
// Geometry.  Workgroup = 512 threads = 8 waves as 2 (rows, wr) x 4 (columns, wc); a wave owns 128 x 64 of the tile = 4 x 2
// accumulators of v_mfma_f32_32x32x16_bf16 / v_mfma_scale_f32_32x32x64_f8f6f4 (128 VGPRs).  A K tile is 128 BYTES of every
// operand row (64 bf16 or 128 e4m3 channels of one tap) staged as FOUR half-tiles of 16 KiB in the order B0, A0, B1, A1:
// A-half h = rows {wr*128 + h*64 .. +63} of both wave rows, B-half h = columns {wc*64 + h*32 .. +31} of all four wave columns.
// Half-tile g = 4 t + kind lives in slot g mod 8 of a 128 KiB ring.  A K tile is four phases, quadrant order (A0,B0) (A0,B1)
// (A1,B1) (A1,B0); each phase reads ONE new register sub-tile from LDS, issues ONE half-tile (two DMA instructions per thread)
// five half-tiles ahead and runs 8 (bf16: 16-deep) or 4 (e4m3: 64-deep) MFMAs per accumulator pair.
//
// Synchronisation (derivation in tools/micro/gemm256_stagger.hip's header and tools/micro/schedule_check.py, which model
// this exact schedule).  Phase = [reads + DMA issue] s_barrier [MFMAs, counted vmcnt] s_barrier, raw barriers, never vmcnt(0)
// inside the loop.  Wave row 1 executes one extra barrier before the loop (row 0 one after it), so one group's LDS reads are
// served while the other's MFMAs run.  RAW: half-tile g is issued in phase g - 5, retired by the counted wait at the end of
// phase g - 3 and first read in phase g - 1: one full barrier after every wave's share has landed.  WAR: g overwrites the
// slot of g - 8, last read three phases before the issue.  Every wave executes the same number of barriers; no spin, flag
// or data-dependent loop.
//
// Operands come through RAW BUFFER DESCRIPTORS (buffer_load_dwordx4 ... lds): a lane's byte offset is computed once; taps
// outside the image and rows past M get an offset beyond num_records, for which the hardware writes zeros to LDS (no zero
// page, no select, no 64-bit address arithmetic); the tile's tap / channel-chunk displacement is a wave-uniform SGPR
// soffset.  The A descriptor's base is moved back by one image row + one pixel so that every tap's displacement is >= 0.
// LDS rows are 128 bytes, two per 256-byte bank row; chunk position p of row i holds the row's chunk p ^ ((i >> 1) & 7)
// (swizzle on the SOURCE address, same XOR on the fragment read).
//
// Epilogue: the shared gemm16_epilogue (bias / time bias / activation / factors / residual / fp32, bf16 and e4m3 copies /
// fused GroupNorm partials / split-K partial store) once per 32-row block of the wave, through a wave-private LDS tile laid
// over the drained ring.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "gemm_epilogue.h"

namespace ddpmir {
namespace {

constexpr int HT_BYTES = 16384;                 // one half-tile: 128 rows x 128 bytes
constexpr int RING_BYTES = 8 * HT_BYTES;
constexpr unsigned OOB = 0x80000000u;           // voffset of a lane that must read zeros (+ soffset < 2^31 never wraps)
constexpr int D = 5;                            // half-tiles between issue and the phase that first needs the item

template <bool F8>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const IGemm p, const int ntn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wr = wave_s >> 2, wc = wave_s & 3;
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware, bijective remap: the workgroups of one XCD get consecutive tiles (neighbours share A rows / 3x3 halo rows)
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int tm = wg / ntn, tn = wg - tm * ntn;
  const int m0 = tm << 8, n0 = tn << 8;
  const int M = p.B * p.H * p.W, HW = p.H * p.W;
  constexpr int EB = F8 ? 1 : 2;                         // bytes per K element of the main segment
  const int nit_main = 9 * ((p.Cin * EB) >> 7);          // K tiles of the nine taps; the second segment (bf16 1x1) follows
  const int KTall = nit_main + ((p.Cinb * 2) >> 7);
  const int nsplit = gridDim.y, sk = blockIdx.y;
  const int t0 = (int)((unsigned)(KTall * sk) / (unsigned)nsplit);               // (KTall * nsplit < 2^31)
  const int KT = (int)((unsigned)(KTall * (sk + 1)) / (unsigned)nsplit) - t0, G = 4 * KT;   // host: KT >= 2
  const unsigned lds0 = (unsigned)(size_t)smem;

  // ---- DMA sources.  Slot s = j*512 + tid of a half-tile: LDS row i = s >> 3 = 64 j + (tid >> 3), 16-byte position s & 7 holds the
  // row's chunk (s & 7) ^ ((i >> 1) & 7) -- the same chunk for a thread's four slots, and their rows are 64 apart:
  //   A slot k = 2 j + h: pixel m_base + 64 k;     B slot (j, h): weight row col_base + 128 j + 32 h
  // so one byte offset per operand and wave-uniform multiples of the row stride describe them all.
  const int ir = tid >> 3;
  const unsigned chunk16 = (unsigned)(((tid & 7) ^ ((ir >> 1) & 7)) << 4);
  const int m_base = m0 + ir;
  const unsigned ldaB = (unsigned)p.lda16 * EB, cinB = (unsigned)p.Cin * EB;
  const unsigned shiftA = (unsigned)(p.W + 1) * ldaB;                      // descriptor base = A - one row - one pixel
  const unsigned offA = (unsigned)m_base * ldaB + chunk16;
  const unsigned offA2 = (unsigned)m_base * (unsigned)p.lda16b * 2u + chunk16;
  const int col_base = n0 + (ir >> 5) * 64 + (ir & 31);
  const unsigned offB = (unsigned)col_base * cinB + chunk16;
  const unsigned offB2 = (unsigned)col_base * (unsigned)p.Cinb * 2u + chunk16;
  // which of the nine taps of slot k's pixel lie inside its image (bit tap; 0 for rows past M: those read zeros everywhere)
  unsigned tapmask[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int m = m_base + 64 * k;
    unsigned mk = 0;
    if (m < M) {
      const int b = (int)p.fd_hw.div((unsigned)m), rem = m - b * HW;
      const int y = (int)p.fd_w.div((unsigned)rem), x = rem - y * p.W;
      const unsigned cx = (x >= 1 ? 1u : 0u) | 2u | (x + 1 < p.W ? 4u : 0u);       // columns dx = -1, 0, +1
      mk = (y >= 1 ? cx : 0u) | (cx << 3) | (y + 1 < p.H ? cx << 6 : 0u);           // rows dy = -1, 0, +1
    }
    tapmask[k] = mk;
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.A16)) - shiftA, 0, (int)((size_t)M * ldaB + shiftA), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.Wt16)), 0, (int)((size_t)9 * p.Cout * cinB), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.Cinb ? p.A16b : p.A16)), 0, p.Cinb ? (int)((size_t)M * p.lda16b * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.Cinb ? p.Wt16b : p.Wt16)), 0, p.Cinb ? (int)((size_t)p.Cout * p.Cinb * 2) : 0, 0x00020000);

  // kind 0 = B0, 1 = A0, 2 = B1, 3 = A1 of the split's K tile `tl` into ring slot `slot`: exactly two DMA instructions per thread
  auto issue = [&](auto kind_c, int tl, int slot) {
    constexpr int KIND = decltype(kind_c)::value, h = KIND >> 1;
    constexpr bool isA = KIND & 1;
    const int t = t0 + tl;                               // K tile of the whole contraction (wave-uniform)
    const unsigned ldsb = lds0 + slot * HT_BYTES + wave_s * 1024;
    if (t >= nit_main) {                                 // second K segment: the block's 1x1 shortcut (bf16 in both modes)
      const unsigned c0 = (unsigned)(t - nit_main) * 128u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (isA) {
          const unsigned v = (tapmask[2 * j + h] & 16u) ? offA2 : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr_t)(size_t)(ldsb + j * 8192), 16, v,
                                                   c0 + (unsigned)(64 * (2 * j + h)) * (unsigned)p.lda16b * 2u, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr_t)(size_t)(ldsb + j * 8192), 16, offB2,
                                                   c0 + (unsigned)(128 * j + 32 * h) * (unsigned)p.Cinb * 2u, 0, 0);
        }
      }
      return;
    }
    // channel chunk outer, tap inner: the nine taps of one chunk re-read the same A lines back to back (L2 hits)
    const int ck = t / 9, tap = t - ck * 9;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (isA) {
        const unsigned v = ((tapmask[2 * j + h] >> tap) & 1u) ? offA : OOB;
        const int dy3 = tap / 3, dx3 = tap - dy3 * 3;    // (dy + 1, dx + 1): the descriptor's base sits one row + one pixel before A
        const unsigned so = (unsigned)ck * 128u + (unsigned)(64 * (2 * j + h) + dy3 * p.W + dx3) * ldaB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(size_t)(ldsb + j * 8192), 16, v, so, 0, 0);
      } else {
        const unsigned so = (unsigned)tap * (unsigned)p.Cout * cinB + (unsigned)ck * 128u + (unsigned)(128 * j + 32 * h) * cinB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(size_t)(ldsb + j * 8192), 16, offB, so, 0, 0);
      }
    }
  };

  // ---- fragment reads: lane (lr, lh) reads row lr of a 32-row sub-tile; key = (lr >> 1) & 7
  const unsigned char* fa = smem + (wr * 64 + lr) * 128;
  const unsigned char* fb = smem + (wc * 32 + lr) * 128;
  // (F8: the second K segment stays bf16 -- its tiles use the bf16 chunk order and MFMA; its weights carry the fp8 weights'
  //  power-of-two scale so that one acc_scale undoes both, as in gemm16_kernel)
  int coff[4], coff16[F8 ? 4 : 1];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c16 = (((lh ^ (lr >> 1)) & 7) ^ (2 * ks)) << 4;
    coff[ks] = F8 ? ((4 * (ks >> 1) + 2 * lh + (ks & 1)) ^ ((lr >> 1) & 7)) << 4      // a[..][2 k8], a[..][2 k8 + 1] = one 32-byte fragment
                  : c16;
    if constexpr (F8) coff16[ks] = c16;
  }

  bf16x8 a[2][4], b[2][4];
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // s2: the tile being read belongs to the second K segment (wave-uniform; only ever true in the F8 kernel, where it selects the bf16 chunk order)
  auto rdA = [&](int slot, bool s2) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        int co = coff[ks];
        if constexpr (F8) co = s2 ? coff16[ks] : co;
        a[mt][ks] = *reinterpret_cast<const bf16x8*>(fa + slot * HT_BYTES + mt * 4096 + co);
      }
  };
  auto rdB = [&](int slot, bf16x8 (&dst)[4], bool s2) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      int co = coff[ks];
      if constexpr (F8) co = s2 ? coff16[ks] : co;
      dst[ks] = *reinterpret_cast<const bf16x8*>(fb + slot * HT_BYTES + co);
    }
  };
  auto bar = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // phase Q8 = 4 * (tile parity) + quadrant of the pair of K tiles (2 u, 2 u + 1)
  // S2 (compile time): the phase's K tile belongs to the second segment (bf16 tiles inside the F8 kernel; always false in the bf16 kernel)
  auto phase = [&](auto q_c, auto s2_c, int u) {
    constexpr int Q8 = decltype(q_c)::value, PAR = Q8 >> 2, q = Q8 & 3, X = PAR, Y = PAR ^ 1;
    constexpr bool s2 = decltype(s2_c)::value;
    const int t = 2 * u + PAR, ph = 4 * t + q;
    // ---------------- read section
    const bool s2n = F8 && t0 + t + 1 >= nit_main;      // the NEXT tile's kind only selects the chunk order of its B0 read
    if constexpr (q == 0) rdA(4 * PAR + 1, s2);
    if constexpr (q == 1) rdB(4 * PAR + 2, b[Y], s2);
    if constexpr (q == 2) rdA(4 * PAR + 3, s2);
    if constexpr (q == 3) { if (t + 1 < KT) rdB(4 * (PAR ^ 1), b[Y], s2n); }
    __builtin_amdgcn_sched_barrier(0);
    if (ph + D < G) issue(std::integral_constant<int, (q + D) & 3>{}, t + (q + D) / 4, (Q8 + D) & 7);
    bar();
    // ---------------- multiply section
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
    constexpr int MI = (q >= 2) ? 2 : 0, NI = (q == 1 || q == 2) ? 1 : 0, BS = (q == 0 || q == 3) ? X : Y;
    if constexpr (F8 && !s2) {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      typedef int i32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int k8 = 0; k8 < 2; ++k8) {
        const i32x8 b8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, b[BS][2 * k8]), __builtin_bit_cast(i32x4, b[BS][2 * k8 + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const i32x8 a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, a[mt][2 * k8]), __builtin_bit_cast(i32x4, a[mt][2 * k8 + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
          acc[MI + mt][NI] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[MI + mt][NI], 0, 0, 0, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[MI + mt][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], b[BS][ks], acc[MI + mt][NI], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    // in flight after this wait: the half-tiles ph + 4 .. min(ph + D, G - 1)
    if (ph + 5 < G) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (ph + 4 < G) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
  };

  // ---------------- prologue: half-tiles 0..4 in flight, 0..2 landed, B0 of tile 0 in registers
  issue(std::integral_constant<int, 0>{}, 0, 0);
  issue(std::integral_constant<int, 1>{}, 0, 1);
  issue(std::integral_constant<int, 2>{}, 0, 2);
  issue(std::integral_constant<int, 3>{}, 0, 3);
  issue(std::integral_constant<int, 0>{}, 1, 4);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // half-tiles 0, 1, 2 landed; 3, 4 in flight
  bar();
  rdB(0, b[0], F8 && t0 >= nit_main);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slot 0 is refilled (half-tile 8) before group 1's first multiply section
  if (wr == 1) bar();
  using no_t = std::integral_constant<bool, false>;
  using yes_t = std::integral_constant<bool, true>;
  // F8 with a second segment: the main tiles of this launch (host: an even number, one split) run the e4m3 phases, the rest the bf16 ones
  const int KT1 = (F8 && p.Cinb) ? nit_main : KT;
  for (int u = 0; u < (KT1 >> 1); ++u) {
    phase(std::integral_constant<int, 0>{}, no_t{}, u);
    phase(std::integral_constant<int, 1>{}, no_t{}, u);
    phase(std::integral_constant<int, 2>{}, no_t{}, u);
    phase(std::integral_constant<int, 3>{}, no_t{}, u);
    phase(std::integral_constant<int, 4>{}, no_t{}, u);
    phase(std::integral_constant<int, 5>{}, no_t{}, u);
    phase(std::integral_constant<int, 6>{}, no_t{}, u);
    phase(std::integral_constant<int, 7>{}, no_t{}, u);
  }
  if constexpr (F8) {
    for (int u = (KT1 >> 1); u < (KT >> 1); ++u) {
      phase(std::integral_constant<int, 0>{}, yes_t{}, u);
      phase(std::integral_constant<int, 1>{}, yes_t{}, u);
      phase(std::integral_constant<int, 2>{}, yes_t{}, u);
      phase(std::integral_constant<int, 3>{}, yes_t{}, u);
      phase(std::integral_constant<int, 4>{}, yes_t{}, u);
      phase(std::integral_constant<int, 5>{}, yes_t{}, u);
      phase(std::integral_constant<int, 6>{}, yes_t{}, u);
      phase(std::integral_constant<int, 7>{}, yes_t{}, u);
    }
  }
  if ((KT & 1) && KT1 == KT) {                           // odd tile count (never with an F8 second segment): the last tile has even parity again
    const int u = KT >> 1;
    phase(std::integral_constant<int, 0>{}, no_t{}, u);
    phase(std::integral_constant<int, 1>{}, no_t{}, u);
    phase(std::integral_constant<int, 2>{}, no_t{}, u);
    phase(std::integral_constant<int, 3>{}, no_t{}, u);
  }
  if (wr == 0) bar();

  // ---------------- epilogue.  Every DMA has landed and every wave is past its last fragment read (vmcnt(0) + barrier of the
  // last phase, group 0's extra barrier): wave w lays its epilogue tile over ring bytes [16 KiB w, +16 KiB).
  // acc[i][j][r] = row wr*128 + i*32 + (r&3) + 8 (r>>2) + 4 lh, column wc*64 + j*32 + lr
  if constexpr (F8) {   // undo the per-tensor weight scale (a power of two)
    const float sc = p.acc_scale;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
  }
  static_assert(epi_ctw<2>() * 4 <= HT_BYTES, "epilogue tile must fit the wave's 16 KiB of the ring");
  float* ct = reinterpret_cast<float*>(smem + wave_s * HT_BYTES);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    gemm16_epilogue<2, false>(p, acc[i], ct, m0 + wr * 128 + i * 32, n0 + wc * 64, M, HW, nsplit, sk);
}

}  // namespace

// Shapes the eight-wave tile takes (everything else stays on gemm16_kernel): 3x3, whole 256-column tiles, whole 128-byte
// channel chunks (of both K segments), operands addressable with 32-bit offsets.
bool gemm256_supported(const IGemm& p) {
  const int eb = p.f8 ? 1 : 2;
  const int64_t M = (int64_t)p.B * p.H * p.W;
  const size_t lim = (size_t)1 << 31;
  const int ktall = 9 * ((p.Cin * eb) >> 7) + ((p.Cinb * 2) >> 7);
  return p.taps == 9 && p.Cout >= 256 && (p.Cout % 256) == 0 && ((p.Cin * eb) % 128) == 0 && p.Cin > 0 && M >= 256 && ktall >= 4 &&
         !p.pk_q && !p.tap_pack && (!p.Cinb || ((p.Cinb % 64) == 0 && p.A16b && p.Wt16b && (size_t)M * p.lda16b * 2 < lim && (size_t)p.Cout * p.Cinb * 2 < lim)) &&
         (size_t)M * p.lda16 * eb + (size_t)(p.W + 1) * p.lda16 * eb < lim && (size_t)p.taps * p.Cout * p.Cin * eb < lim &&
         (!p.gn_sums || ((p.H * p.W) % 32) == 0) &&
         (!(p.f8 && p.Cinb) || ((9 * (p.Cin >> 7)) % 2 == 0 && ((p.Cinb >> 6) % 2) == 0));   // e4m3 + bf16 segment: whole tile pairs of each kind
}

// split-K factor for `tiles` output tiles and `ktall` K tiles (1 = none): fill the chip, keep >= 4 K tiles per split
// (measured, tools/gemm256_ab.py: a split that leaves >= 16 K tiles wins 1.08-1.3x over the 128-row kernel, a shorter one loses to
//  the fp32 round trip of the partial sums; `forced` keeps the old >= 4 so that the tests reach the split path with small shapes)
int gemm256_splits(const IGemm& p, int64_t tiles, int ktall, int n_cu, bool forced) {
  const int64_t M = (int64_t)p.B * p.H * p.W;
  const bool linear = p.epi_mask == MASK_NONE && (p.out || p.out16 || p.out8) && !p.pk_q && !p.sel_C && !p.gn_sums && p.splitk_ws;
  if (!linear || tiles >= (3 * n_cu) / 4 || (!forced && ktall < 64) || (p.f8 && p.Cinb)) return 1;   // (36 K tiles in two splits measured 0.90-0.92x)
  int ns = (int)std::min<int64_t>(cdiv64(n_cu, tiles), ktall / (forced ? 4 : 16));
  ns = (int)std::min<int64_t>(ns, (int64_t)(p.splitk_ws_bytes / ((size_t)M * p.Cout * sizeof(float))));
  return ns < 1 ? 1 : ns;
}

// (defined in gemm16.hip) the fixed-order fold of split-K slices + the linear epilogue
void launch_splitk_reduce(const IGemm& p, int nsplit, int64_t M, hipStream_t stream);

// launches taken by the tile since the library was loaded (tests assert that a shape meant for it did not silently fall back)
static std::atomic<long long> g_launches{0};
long long gemm256_launch_count() { return g_launches.load(); }

// DDPMIR_GEMM256_LOG=1 (diagnostics, read once): one stderr line per 3x3 launch with Cout >= 256 saying which kernel took it and why
static bool log_enabled() {
  static const bool on = [] { const char* e = getenv("DDPMIR_GEMM256_LOG"); return e && *e && *e != '0'; }();
  return on;
}
static bool declined(const IGemm& p, const char* why) {
  if (log_enabled() && p.taps == 9 && p.Cout >= 256)
    fprintf(stderr, "[gemm256] B=%d %dx%d %d->%d%s%s%s: 128-row kernel (%s)\n", p.B, p.H, p.W, p.Cin, p.Cout, p.f8 ? " e4m3" : "",
            p.Cinb ? " +seg2" : "", p.gn_sums ? " +gn" : "", why);
  return false;
}

bool launch_gemm256(const IGemm& p, hipStream_t stream) {
  if (p.tile256 == 0) return declined(p, "option");
  if (!gemm256_supported(p)) return declined(p, "unsupported shape");
  const int64_t M = (int64_t)p.B * p.H * p.W;
  static int n_cu = 0;
  static PerDeviceOnce once;
  once.run([] {
    int dev = 0;
    hipDeviceProp_t prop;
    DDPMIR_HIP_CHECK(hipGetDevice(&dev));
    DDPMIR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount;
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
  });
  const int ntn = p.Cout >> 8;
  const int64_t tiles = cdiv64(M, 256) * ntn;
  const int eb = p.f8 ? 1 : 2;
  const int ktall = 9 * ((p.Cin * eb) >> 7) + ((p.Cinb * 2) >> 7);
  const int cu = n_cu > 0 ? n_cu : 256;
  const int nsplit = gemm256_splits(p, tiles, ktall, cu, p.tile256 > 0);
  // auto mode: only where the tile fills the chip (with the K split where the epilogue allows one and the split still leaves
  // a long K loop); few-tile launches with a fused epilogue stay on the 128-row kernel, which has twice the tiles
  if (p.tile256 < 0 && tiles * nsplit < (3 * cu) / 4) return declined(p, "too few tiles");
  if (ktall / nsplit < 2) return declined(p, "K too short");
  IGemm q = p;
  q.nmajor = 0; q.tap_pack = 0;
  const dim3 grid((unsigned)tiles, (unsigned)nsplit, 1);
  if (p.f8) hipLaunchKernelGGL((gemm256_kernel<true>), grid, dim3(512), RING_BYTES, stream, q, ntn);
  else hipLaunchKernelGGL((gemm256_kernel<false>), grid, dim3(512), RING_BYTES, stream, q, ntn);
  if (nsplit > 1) launch_splitk_reduce(p, nsplit, M, stream);
  g_launches.fetch_add(1);
  if (log_enabled())
    fprintf(stderr, "[gemm256] B=%d %dx%d %d->%d%s%s%s: 256 tile, %lld tiles x %d splits (%d K tiles)\n", p.B, p.H, p.W, p.Cin, p.Cout,
            p.f8 ? " e4m3" : "", p.Cinb ? " +seg2" : "", p.gn_sums ? " +gn" : "", (long long)tiles, nsplit, ktall);
  return true;
}

}  // namespace ddpmir

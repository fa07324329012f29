// bf16 implicit GEMM with BOTH operands streamed by LDS-DMA (global_load_lds_dwordx4), gfx950.
//
//   out[m][n] = epi( sum_{tap,c} A16[pixel(m)+tap][c] * W16[tap][n][c] )
//
// A16 is a bf16 NHWC tensor that already contains what the GEMM consumes (normalised / activated / masked by
// the producing kernel), so nothing has to pass through VGPRs on the way in: every wave issues 16-byte-per-lane
// direct-to-LDS loads into a stage ring -- two stages in production: the next K chunk is in flight across the raw
// s_barrier while the MFMAs of the current chunk run, and three or four workgroups per CU hide each other's waits
// (a three-stage ring with a counted vmcnt is still instantiable: it lost to the smaller footprint).
// Kernels: gemm16_kernel (generic: 1x1 / 3x3, optional second K segment, fp8 operands, split-K),
// conv3x3_strip_kernel (3x3 with <= 64 output channels at the large levels), one shared epilogue (gemm16_epilogue).
//
// LDS image per stage: A [128 rows][BK] bf16, B [32*NT rows][BK] bf16, unpadded rows of BK*2 bytes (BK = 32 or 64).
// A DMA wave-instruction writes 64 lanes x 16 B linearly, so the bank-conflict fix is an XOR swizzle on the SOURCE
// address: LDS chunk p of row r holds global chunk p ^ key(r), key = (r >> log2(rows per 256-B bank row)) mod
// chunks-per-row, which makes every 16-lane group of a ds_read_b128 hit 16 distinct 16-byte slots; the MFMA operand
// read applies the same XOR (rule: linear destination + permuted source + permuted read).  Rows outside the image (3x3 halo),
// rows past M / Cout and channel chunks past Cin are fetched from a 16-byte zero page.
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "kernels.h"
#include "gemm_epilogue.h"

namespace ddpmir {
namespace {

constexpr int BM = 128;

// BK = K elements per stage (64 or 32).  BK = 32 keeps the workgroup at 32 KiB of ring so four of them share one CU:
// one workgroup's barrier waits, LDS-read latency and epilogue overlap the MFMAs of the others.
// STAGES = 2 (double buffering, one stage in flight per workgroup): measured better than a 3-stage ring with two stages
// in flight because the smaller footprint admits a fourth workgroup per CU at BK = 32 (32 KiB ring, 36 KiB epilogue
// tile) and a second one at BK = 64 -- occupancy hides the load latency better than a deeper ring does (3x3 convs of
// levels 1-3: -3 %, levels 4-5: -11 %, level-0 1x1: -25 %).
//
// F8 = 1: both operands are OCP e4m3 bytes (same LDS geometry: a row of BK*2 bytes now holds 2*BK K elements, a 16-byte
// chunk 16 of them) and the contraction runs on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: twice the bf16
// MFMA rate and half the operand bytes per MAC.  The per-tensor weight scale is undone on the accumulators (acc_scale).
// The 128-column bf16 tile at BK = 32 is compiled for four waves per SIMD (<= 128 VGPRs; the compiler spills two registers
// outside the K loop): its 36 KiB LDS footprint already allows four workgroups per CU, 133 VGPRs only three.  A/B on one
// device (tools/lib_ab_bench.py): forward minus level-0 attention 15.26 -> 14.93 ms, all GEMMs of a forward 6.98 -> 6.85 ms.
template <int NT, int BK, int STAGES, int F8 = 0>
__global__ __launch_bounds__(256, (NT == 4 && BK == 32 && F8 == 0) ? 4 : 1) void gemm16_kernel(const IGemm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWB = BK * 2;                         // bytes per LDS row
  constexpr int CPR = BK / 8;                          // 16-byte chunks per row
  constexpr int EB = F8 ? 1 : 2;                       // bytes per operand element
  constexpr int EPC = 16 / EB;                         // K elements per 16-byte chunk
  constexpr int BKE = CPR * EPC;                       // K elements per stage
  constexpr int RSH = BK == 64 ? 1 : 2;                // rows per 256-byte bank row = 1 << RSH
  constexpr int BROWS = NT * 32 * CPR < 256 ? 256 / CPR : NT * 32;   // every thread issues >= 1 B piece
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BROWS * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int AL = BM * CPR / 256, BL = BROWS * CPR / 256;          // DMA instructions per thread per stage
  constexpr int LOADS = AL + BL;
  constexpr int KS = BK / 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.B * p.H * p.W, HW = p.H * p.W;
  // XCD-aware placement: consecutive M tiles (which share 3x3 halo rows) run on the same XCD / L2
  // Weight-bound shapes (few M tiles, p.nmajor): blockIdx.x is the N tile instead, so the M tiles that stream the
  // SAME weight slice get workgroup ids a multiple of gridDim.x apart = the same XCD, and the slice is fetched into one
  // L2 instead of eight (the activations are small enough to sit in every L2).
  int bx, by;
  if (p.nmajor) { bx = blockIdx.y; by = blockIdx.x; }
  else {
    bx = blockIdx.x; by = blockIdx.y;
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
  }
  const int m0 = bx * BM, n0 = by * (NT * 32);
  const int nck = (p.Cin + BKE - 1) / BKE;
  const int nit_main = p.tap_pack ? (9 + CPR - 1) / CPR : p.taps * nck;
  // (F8: the second K segment -- the block's 1x1 shortcut -- stays bf16: its stages use the bf16 fragment reads and MFMA, and
  //  its weights carry the fp8 weights' power-of-two scale so that one acc_scale undoes both)
  const int nit_all = nit_main + (p.Cinb + BK - 1) / BK;
  const int nsplit = gridDim.z, sk = blockIdx.z;
  const int it0 = (int)((long long)nit_all * sk / nsplit), it1 = (int)((long long)nit_all * (sk + 1) / nsplit);
  const int nit = (GEMM_DBG(p) & 16) ? 0 : it1 - it0;
  const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A16);
  const __bf16* W16 = reinterpret_cast<const __bf16*>(p.Wt16);
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(p.A16);    // F8: the same fields carry e4m3 tensors
  const unsigned char* W8 = reinterpret_cast<const unsigned char*>(p.Wt16);
  const __bf16* A16b = reinterpret_cast<const __bf16*>(p.A16b);
  const __bf16* W16b = reinterpret_cast<const __bf16*>(p.Wt16b);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(p.zero16) +
                              (((blockIdx.x + 3 * blockIdx.y + 5 * blockIdx.z) & 15) * 256 + tid) * 16;   // 16 x 4 KiB windows
  const unsigned lds0 = (unsigned)(size_t)smem;

  // per-thread DMA slots: slot s = j*256 + tid covers LDS row s / CPR, chunk position s % CPR
  int a_y[AL], a_x[AL], a_m[AL];
  bool a_ok[AL];
  size_t a_off[AL];
  int a_chunk[AL];
#pragma unroll
  for (int j = 0; j < AL; ++j) {
    const int row = (j * 256 + tid) / CPR, pos = tid & (CPR - 1);
    const int m = m0 + row;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int b = (int)p.fd_hw.div((unsigned)mm), rem = mm - b * HW;
    a_y[j] = (int)p.fd_w.div((unsigned)rem); a_x[j] = rem - a_y[j] * p.W;
    a_ok[j] = ok;
    a_off[j] = (size_t)mm * p.lda16; a_m[j] = mm;
    a_chunk[j] = pos ^ ((row >> RSH) & (CPR - 1));     // which 8-channel chunk of the row this slot holds
  }
  int b_n[BL], b_chunk[BL];
#pragma unroll
  for (int j = 0; j < BL; ++j) {
    const int row = (j * 256 + tid) / CPR, pos = tid & (CPR - 1);
    b_n[j] = row < NT * 32 ? n0 + row : p.Cout;         // rows past the tile read the zero page
    b_chunk[j] = pos ^ ((row >> RSH) & (CPR - 1));
  }

  auto issue = [&](int itl) {
    const int it = it0 + itl;
    const unsigned sbase2 = lds0 + (itl % STAGES) * STAGE;
    if (it >= nit_main) {   // second K segment: 1x1 operand pair (wave-uniform branch)
      const int c0 = (it - nit_main) * BK;
#pragma unroll
      for (int j = 0; j < AL; ++j) {
        const int c = c0 + 8 * a_chunk[j];
        const bool ok = a_ok[j] && c < p.Cinb;
        const void* g = ok ? (const void*)(A16b + (size_t)a_m[j] * p.lda16b + c) : (const void*)zero;
        __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sbase2 + (j * 256 + wave * 64) * 16), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BL; ++j) {
        const int c = c0 + 8 * b_chunk[j];
        const bool ok = b_n[j] < p.Cout && c < p.Cinb;
        const void* g = ok ? (const void*)(W16b + (size_t)b_n[j] * p.Cinb + c) : (const void*)zero;
        __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sbase2 + A_BYTES + (j * 256 + wave * 64) * 16), 16, 0, 0);
      }
      return;
    }
    if (!F8 && p.tap_pack) {   // 8-channel 3x3 operand: chunk q of the K axis = tap q (wave-uniform branch)
      const unsigned sb = lds0 + (itl % STAGES) * STAGE;
#pragma unroll
      for (int j = 0; j < AL; ++j) {
        const int q = it * CPR + a_chunk[j];
        const int dy = q / 3 - 1, dx = q - (q / 3) * 3 - 1;
        const bool ok = a_ok[j] && q < 9 && (unsigned)(a_y[j] + dy) < (unsigned)p.H && (unsigned)(a_x[j] + dx) < (unsigned)p.W;
        const void* g = ok ? (const void*)(A16 + a_off[j] + (ptrdiff_t)(dy * p.W + dx) * p.lda16) : (const void*)zero;
        __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sb + (j * 256 + wave * 64) * 16), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BL; ++j) {
        const int q = it * CPR + b_chunk[j];
        const bool ok = b_n[j] < p.Cout && q < 9;
        const void* g = ok ? (const void*)(W16 + ((size_t)q * p.Cout + b_n[j]) * 8) : (const void*)zero;
        __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sb + A_BYTES + (j * 256 + wave * 64) * 16), 16, 0, 0);
      }
      return;
    }
    // channel chunk outer, tap inner: the nine taps of one chunk re-read the same few A cache lines back to back (L2 / TCP
    // hits).  Tap-major order streamed the whole channel depth between two uses of a line and re-fetched every tap through
    // the fabric (PMC: 1 096 MB fetched for a 134 MB operand at 256 -> 64, 128 x 128: 181 -> 159 us).
    const int ck = it / p.taps, tap = it - ck * p.taps, c0 = ck * BKE;
    int dy = 0, dx = 0;
    if (p.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
    const unsigned sbase = lds0 + (itl % STAGES) * STAGE;
    if (!(GEMM_DBG(p) & 1))
#pragma unroll
    for (int j = 0; j < AL; ++j) {
      const int c = c0 + EPC * a_chunk[j];
      const bool ok = a_ok[j] && c < p.Cin && (unsigned)(a_y[j] + dy) < (unsigned)p.H && (unsigned)(a_x[j] + dx) < (unsigned)p.W;
      const void* g = !ok ? (const void*)zero
                          : F8 ? (const void*)(A8 + a_off[j] + (ptrdiff_t)(dy * p.W + dx) * p.lda16 + c)
                               : (const void*)(A16 + a_off[j] + (ptrdiff_t)(dy * p.W + dx) * p.lda16 + c);
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sbase + (j * 256 + wave * 64) * 16), 16, 0, 0);
    }
    if (!(GEMM_DBG(p) & 2))
#pragma unroll
    for (int j = 0; j < BL; ++j) {
      const int c = c0 + EPC * b_chunk[j];
      const bool ok = b_n[j] < p.Cout && c < p.Cin;
      const void* g = !ok ? (const void*)zero
                          : F8 ? (const void*)(W8 + ((size_t)tap * p.Cout + b_n[j]) * p.Cin + c)
                               : (const void*)(W16 + ((size_t)tap * p.Cout + b_n[j]) * p.Cin + c);
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sbase + A_BYTES + (j * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const int arow = wave * 32 + lr;
  // b rows: nt*32 + lr -> the key only depends on lr because 32 is a multiple of (CPR << RSH)
  const int a_key = (arow >> RSH) & (CPR - 1), b_key = (lr >> RSH) & (CPR - 1);

  constexpr int AHEAD = STAGES - 1;                     // stages in flight: STAGES - 1
#pragma unroll
  for (int i = 0; i < AHEAD; ++i)
    if (i < nit) issue(i);
  for (int it = 0; it < nit; ++it) {
    // stage `it` has landed once at most the younger stages of THIS wave are still in flight (AHEAD - 1 of them, fewer at the
    // end of the loop); the barrier extends that to every wave's share and also retires everybody's reads of the buffer
    // that is refilled next
    const int younger = nit - 1 - it < AHEAD - 1 ? nit - 1 - it : AHEAD - 1;
    if (AHEAD > 2 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
    else if (AHEAD > 1 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + AHEAD < nit) issue(it + AHEAD);
    const unsigned char* st = smem + (it % STAGES) * STAGE;
    if (F8 && it0 + it < nit_main) {
      // one 32x32x64 MFMA per 64 K bytes of the row: lane half lh holds k = 32 lh .. 32 lh + 31 = chunks 2 lh, 2 lh + 1
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      constexpr int KS8 = ROWB / 64;
      i32x8 a8[KS8], b8[KS8][NT];
#pragma unroll
      for (int ks = 0; ks < KS8; ++ks) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(st + arow * ROWB + (((4 * ks + 2 * lh) ^ a_key) << 4));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(st + arow * ROWB + (((4 * ks + 2 * lh + 1) ^ a_key) << 4));
        a8[ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const unsigned char* br = st + A_BYTES + (nt * 32 + lr) * ROWB;
          const i32x4 blo = *reinterpret_cast<const i32x4*>(br + (((4 * ks + 2 * lh) ^ b_key) << 4));
          const i32x4 bhi = *reinterpret_cast<const i32x4*>(br + (((4 * ks + 2 * lh + 1) ^ b_key) << 4));
          b8[ks][nt] = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS8; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[ks], b8[ks][nt], acc[nt], 0, 0, 0, 0, 0, 0);
      continue;
    }
    // all operand fragments of the stage first (one LDS round trip), then the MFMAs back to back
    bf16x8 af[KS], bfr[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      af[ks] = *reinterpret_cast<const bf16x8*>(st + arow * ROWB + (((2 * ks + lh) ^ a_key) << 4));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bfr[ks][nt] = *reinterpret_cast<const bf16x8*>(st + A_BYTES + (nt * 32 + lr) * ROWB + (((2 * ks + lh) ^ b_key) << 4));
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfr[ks][nt], acc[nt], 0, 0, 0);
  }

  if constexpr (F8) {   // undo the per-tensor weight scale (a power of two)
    const float sc = p.acc_scale;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] *= sc;
  }
  if constexpr (NT <= 2) {
    if (p.epi_direct) { gemm16_epilogue_direct_any<NT>(p, acc, m0 + wave * 32, n0, M, HW); return; }
  }
  gemm16_epilogue<NT>(p, acc, reinterpret_cast<float*>(smem) + wave * epi_ctw<NT>(), m0 + wave * 32, n0, M, HW, nsplit, sk);
}

// 3x3 convs with few output channels (Cout <= 64: the full-resolution levels) are bound by the L2 -> LDS stream of their
// operand tiles, not by the matrix pipe: per 128 x 64 tile and 64-channel tap the generic kernel moves 16 KiB of A + 8 KiB
// of B for 128 * 64 * 64 MACs.  But the three taps of one kernel row read the SAME pixels shifted by one: a tile is 128
// consecutive pixels in NHWC order, so tap dx is the A tile moved by one LDS row.  This kernel stages, per (32-channel
// chunk, kernel row dy), ONE strip of 130 pixel rows (pixel m0 - 1 .. m0 + 128) and the three taps' weights, and reads the A
// fragments of tap dx at row offset dx; lanes whose pixel sits on the left / right image border get a zero fragment for
// dx = -1 / +1 (the shifted row belongs to the neighbouring image row there).  A traffic drops 3x (B unchanged): 216 KiB ->
// 122 KiB per tile at Cin = 64.  Same accumulation order as the generic kernel at BK = 32 (chunk outer, taps 0..8 inner).
//
// BUF = 1: the DMA loads go through raw buffer descriptors (buffer_load_dwordx4 ... lds).  The generic addressing spends ~16
// VALU instructions per MFMA on 64-bit address arithmetic, bounds tests and the zero-page select of every load (PMC,
// profiles/r03_gemm_pmc.txt: 74.9 M VALU instructions for 4.7 M MFMAs: the SIMD's vector issue, not the matrix pipe or the
// memory system, was the busiest resource).  With descriptors a lane's byte offset for each kernel row is precomputed ONCE
// (out-of-image rows and rows past M get an offset beyond num_records: the hardware writes zeros to LDS, no zero page), the
// stage's channel / tap displacement is a wave-uniform SGPR soffset, and a load is issued without any vector arithmetic.
// Needs Cin % 32 == 0 (no channel tail inside a stage) and operands below 2 GiB.
constexpr unsigned BUF_OOB = 0x80000000u;     // voffset of a lane that must read zeros (+ soffset < 2^31 never wraps)

// (three workgroups per CU = three waves per SIMD: the register allocation is capped at 168 VGPRs so that the LDS footprint, not
//  an epilogue's address arithmetic, decides the occupancy)
template <int NT, bool BUF = false>
__global__ __launch_bounds__(256, 3) void conv3x3_strip_kernel(const IGemm p, const int gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BK = 32, ROWB = 64, CPR = 4, RSH = 2, KS = 2, STAGES = 2;
  constexpr int AR = BUF ? BM + 16 : BM + 8;            // strip rows allocated (130 used): row r = pixel m0 - 1 + r
                                                        // (BUF: 144, so that wave 0's whole third piece -- 16 rows -- lands inside the strip)
  constexpr int A_BYTES = AR * ROWB;
  constexpr int BROWS = 64;                             // weight rows per tap (>= NT * 32; 64 x 4 chunks = one slot per thread)
  constexpr int BT_BYTES = BROWS * ROWB, STAGE = A_BYTES + 3 * BT_BYTES;
  constexpr int AL = 3;                                 // 130 rows x 4 chunks = 520 slots: two full rounds + 8 threads
  static_assert(NT * 32 <= BROWS, "strip kernel: Cout tile <= 64");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.B * p.H * p.W, HW = p.H * p.W;
  // gx = number of M tiles.  The grid may be smaller (launch_strip: three workgroups per CU): a workgroup then walks tiles
  // blockIdx.x, blockIdx.x + gridDim.x, ... -- the ~16 000 workgroup launches of a full-resolution conv cost more than the
  // tiles' own prologues (64 -> 64 at 512 x 512, B = 8: half of the launch's 377 us did not depend on K)
  for (int tile = blockIdx.x; tile < gx; tile += gridDim.x) {
  int bx = tile, by = blockIdx.y;
  if ((gx & 7) == 0) bx = (bx & 7) * (gx >> 3) + (bx >> 3);   // neighbouring tiles (shared halo rows) on one XCD
  const int m0 = bx * BM, n0 = by * (NT * 32);
  const int nck = (p.Cin + BK - 1) / BK;
  const int nit_main = 3 * nck;
  const int nit_all = nit_main + (p.Cinb + BK - 1) / BK;
  const int nsplit = gridDim.z, sk = blockIdx.z;
  const int it0 = (int)((long long)nit_all * sk / nsplit), it1 = (int)((long long)nit_all * (sk + 1) / nsplit);
  const int nit = it1 - it0;
  const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A16);
  const __bf16* W16 = reinterpret_cast<const __bf16*>(p.Wt16);
  const __bf16* A16b = reinterpret_cast<const __bf16*>(p.A16b);
  const __bf16* W16b = reinterpret_cast<const __bf16*>(p.Wt16b);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(p.zero16) +
                              (((blockIdx.x + 3 * blockIdx.y + 5 * blockIdx.z) & 15) * 256 + tid) * 16;
  const unsigned lds0 = (unsigned)(size_t)smem;

  // A slots: s = j * 256 + tid -> strip row s / 4 (pixel m0 - 1 + row), chunk position s % 4
  int a_y[AL], a_m[AL], a_chunk[AL];
  bool a_use[AL], a_ok[AL];
#pragma unroll
  for (int j = 0; j < AL; ++j) {
    const int row = (j * 256 + tid) >> 2, pos = tid & 3;
    const int m = m0 - 1 + row;
    a_use[j] = row < BM + 2;
    a_ok[j] = a_use[j] && m >= 0 && m < M;
    const int mm = a_ok[j] ? m : 0;
    const int b = (int)p.fd_hw.div((unsigned)mm), rem = mm - b * HW;
    a_y[j] = (int)p.fd_w.div((unsigned)rem);
    a_m[j] = mm;
    a_chunk[j] = pos ^ ((row >> RSH) & (CPR - 1));
  }
  // B slots: slot j = tap dx = j - 1 of the current kernel row, weight row tid / 4, chunk position tid % 4
  const int b_row = tid >> 2;
  const int b_n = b_row < NT * 32 ? n0 + b_row : p.Cout;       // rows past the tile read the zero page
  const int b_chunk = (tid & 3) ^ ((b_row >> RSH) & (CPR - 1));

  // BUF: per-lane byte offsets, one per kernel row (dy = -1, 0, +1), computed once
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  unsigned av0[AL], av1[AL], av2[AL], avb[AL], bv = BUF_OOB, bvb = BUF_OOB;
  __amdgpu_buffer_rsrc_t rsA, rsB, rsAb, rsBb;
  if constexpr (BUF) {
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A16), 0, (int)((size_t)M * p.lda16 * 2), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W16), 0, (int)((size_t)9 * p.Cout * p.Cin * 2), 0x00020000);
    rsAb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Cinb ? A16b : A16), 0, p.Cinb ? (int)((size_t)M * p.lda16b * 2) : 0, 0x00020000);
    rsBb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Cinb ? W16b : W16), 0, p.Cinb ? (int)((size_t)p.Cout * p.Cinb * 2) : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < AL; ++j) {
      const unsigned base = ((unsigned)a_m[j] * (unsigned)p.lda16 + 8u * a_chunk[j]) * 2u;
      const unsigned rowb = (unsigned)p.W * (unsigned)p.lda16 * 2u;
      av0[j] = (a_ok[j] && a_y[j] >= 1) ? base - rowb : BUF_OOB;
      av1[j] = a_ok[j] ? base : BUF_OOB;
      av2[j] = (a_ok[j] && a_y[j] + 1 < p.H) ? base + rowb : BUF_OOB;
      avb[j] = a_ok[j] ? ((unsigned)a_m[j] * (unsigned)p.lda16b + 8u * a_chunk[j]) * 2u : BUF_OOB;
    }
    if (b_n < p.Cout) {
      bv = ((unsigned)b_n * (unsigned)p.Cin + 8u * b_chunk) * 2u;
      bvb = ((unsigned)b_n * (unsigned)p.Cinb + 8u * b_chunk) * 2u;
    }
  }

  auto issue = [&](int itl) {
    const int it = it0 + itl;
    const unsigned sb = lds0 + (itl % STAGES) * STAGE;
    if constexpr (BUF) {
      // (wave_s: the wave id as an SGPR, so the LDS base of every piece is scalar arithmetic.  Pieces 0 and 1 cover strip rows
      //  0 .. 127; piece 2 covers rows 128 .. 143 and is wave 0's alone -- lanes past row 129 carry BUF_OOB and store zeros)
      const unsigned sbw = sb + wave_s * 1024;
      if (it >= nit_main) {   // second K segment (the block's 1x1 shortcut): centre position only; Cinb may end inside the stage
        const int c0 = (it - nit_main) * BK;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
          if (j == 2 && wave_s != 0) continue;
          const unsigned v = (c0 + 8 * a_chunk[j] < p.Cinb) ? avb[j] : BUF_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsAb, (lds_ptr_t)(size_t)(sbw + j * 4096), 16, v, (unsigned)(c0 * 2), 0, 0);
        }
        const unsigned vb = (c0 + 8 * b_chunk < p.Cinb) ? bvb : BUF_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBb, (lds_ptr_t)(size_t)(sbw + A_BYTES + BT_BYTES), 16, vb, (unsigned)(c0 * 2), 0, 0);
        return;
      }
      const int ck = it / 3, dyi = it - ck * 3;
      const unsigned soA = (unsigned)(ck * BK * 2);
#pragma unroll
      for (int j = 0; j < AL; ++j) {
        if (j == 2 && wave_s != 0) continue;
        const unsigned v = dyi == 0 ? av0[j] : (dyi == 1 ? av1[j] : av2[j]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(size_t)(sbw + j * 4096), 16, v, soA, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned soB = (unsigned)((((dyi * 3 + j) * p.Cout) * p.Cin + ck * BK) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(size_t)(sbw + A_BYTES + j * BT_BYTES), 16, bv, soB, 0, 0);
      }
      return;
    }
    if (it >= nit_main) {   // second K segment (the block's 1x1 shortcut): centre position only
      const int c0 = (it - nit_main) * BK;
#pragma unroll
      for (int j = 0; j < AL; ++j) {
        if (!a_use[j]) continue;
        const int c = c0 + 8 * a_chunk[j];
        const bool ok = a_ok[j] && c < p.Cinb;
        const void* g = ok ? (const void*)(A16b + (size_t)a_m[j] * p.lda16b + c) : (const void*)zero;
        __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sb + (j * 256 + wave * 64) * 16), 16, 0, 0);
      }
      {
        const int c = c0 + 8 * b_chunk;
        const bool ok = b_n < p.Cout && c < p.Cinb;
        const void* g = ok ? (const void*)(W16b + (size_t)b_n * p.Cinb + c) : (const void*)zero;
        __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sb + A_BYTES + BT_BYTES + (wave * 64) * 16), 16, 0, 0);
      }
      return;
    }
    const int ck = it / 3, dyi = it - ck * 3, dy = dyi - 1, c0 = ck * BK;
#pragma unroll
    for (int j = 0; j < AL; ++j) {
      if (!a_use[j]) continue;
      const int c = c0 + 8 * a_chunk[j];
      const bool ok = a_ok[j] && c < p.Cin && (unsigned)(a_y[j] + dy) < (unsigned)p.H;
      const void* g = ok ? (const void*)(A16 + ((ptrdiff_t)a_m[j] + (ptrdiff_t)dy * p.W) * p.lda16 + c) : (const void*)zero;
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sb + (j * 256 + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = c0 + 8 * b_chunk;
      const bool ok = b_n < p.Cout && c < p.Cin;
      const void* g = ok ? (const void*)(W16 + ((size_t)(dyi * 3 + j) * p.Cout + b_n) * p.Cin + c) : (const void*)zero;
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(sb + A_BYTES + j * BT_BYTES + (wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const int arow = wave * 32 + lr;                      // the lane's pixel of the tile = strip row arow + 1
  const int px = (m0 + arow) - (int)p.fd_w.div((unsigned)(m0 + arow)) * p.W;   // its column in the image (m % W == (m % HW) % W): taps that leave the image row are zero
  const bool x_ok[3] = {px > 0, true, px + 1 < p.W};
  const int b_key = (lr >> RSH) & (CPR - 1);
  bf16x8 zero8;
#pragma unroll
  for (int j = 0; j < 8; ++j) zero8[j] = (__bf16)0.f;

  // only the waves whose 32 pixels touch the left or right image border need the zero fragments (2 of W / 32 waves per image row):
  // everybody else runs the tap loop without the per-fragment selects
  const bool wave_border = BUF ? (__any(!x_ok[0] || !x_ok[2]) != 0) : true;
  auto taps = [&](const unsigned char* st, const bool seg2, auto masked) {
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      if (seg2 && dxi != 1) continue;
      const int row = arow + dxi, a_key = (row >> RSH) & (CPR - 1);
      bf16x8 af[KS], bfr[KS][NT];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        af[ks] = *reinterpret_cast<const bf16x8*>(st + row * ROWB + (((2 * ks + lh) ^ a_key) << 4));
        if (decltype(masked)::value && !seg2 && !x_ok[dxi]) af[ks] = zero8;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bfr[ks][nt] = *reinterpret_cast<const bf16x8*>(st + A_BYTES + dxi * BT_BYTES + (nt * 32 + lr) * ROWB + (((2 * ks + lh) ^ b_key) << 4));
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfr[ks][nt], acc[nt], 0, 0, 0);
    }
  };
  if (nit > 0) issue(0);
  for (int it = 0; it < nit; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < nit) issue(it + 1);
    const unsigned char* st = smem + (it % STAGES) * STAGE;
    const bool seg2 = it0 + it >= nit_main;
    if (wave_border) taps(st, seg2, std::true_type());
    else taps(st, seg2, std::false_type());
  }
  if (p.epi_direct) {
    gemm16_epilogue_direct_any<NT>(p, acc, m0 + wave * 32, n0, M, HW);
    if (tile + (int)gridDim.x < gx) __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read before the next tile's first DMA lands
  } else {
  gemm16_epilogue<NT>(p, acc, reinterpret_cast<float*>(smem) + wave * epi_ctw<NT>(), m0 + wave * 32, n0, M, HW, nsplit, sk);
  if (tile + (int)gridDim.x < gx) __syncthreads();   // the epilogue's wave tiles alias the stage ring the next tile refills
  }
  }
}

// ------------------------------------------------------------------------------------------------ weights-stationary 64 -> 64
// EXPERIMENT (round 3, -DDDPMIR_EXPERIMENTS builds only, DDPMIR_WS64=1): measured SLOWER than the strip kernel and not shipped.
// 64 -> 64 at 512 x 512, B = 8: 511 us against 337 us.  Dissection with the dbg switches (tools/ws_dissect2.py,
// profiles/r03_ws64_dissection.txt), per 256-pixel tile: 15.9 us = 8.4 us epilogue + ~2 us K loop (loads and MFMAs overlap:
// either alone costs the same) + 5.4 us that remain with loads, MFMAs, barriers, address set-up AND epilogue all switched off.
// With one 160-KiB workgroup per CU nothing overlaps the epilogue's 98 KiB of stores or any other serial phase, whereas the
// strip kernel's three co-resident workgroups hide each other's; the design would need a second, epilogue-only wave group
// fed through LDS, for which the resident weights leave no room.  Kept for the record of what was tried.
#ifdef DDPMIR_EXPERIMENTS
// The 3x3 convs with 64 input and 64 output channels (conv2 / conv_out of the full-resolution blocks: at 512 x 512, B = 8 a
// third of all GEMM time) spend their time WAITING: per 128-pixel tile the strip kernel runs 72 MFMAs per wave (1.1 us of matrix
// pipe) inside 17.7 us, six stage round trips of ~2.9 us each, and 12 of the 20 KB every stage fetches are the same 74 KB of
// weights, fetched again by each of 16 384 workgroups (PMC, profiles/r03_gemm_pmc.txt: 50 % of the wave cycles parked on
// s_waitcnt / s_barrier, matrix pipe 19 % busy).  This kernel turns that around:
//   * one PERSISTENT workgroup per CU walks a contiguous range of 256-pixel tiles (its own halo rows stay in its XCD's L2);
//   * the whole weight tensor lives in LDS for the workgroup's lifetime (9 x 64 x 64 bf16 = 72 KiB, fetched once);
//   * only the activation strips stream: one 258-row strip per (32-channel chunk, kernel row) = 16.5 KiB per stage, through
//     a THREE-buffer ring with two stages in flight across the raw s_barrier (counted s_waitcnt vmcnt, descriptor-addressed
//     buffer_load ... lds with precomputed per-lane offsets: no vector arithmetic per load) -- and the ring keeps running
//     across tile boundaries, so a tile's first strips arrive while the previous tile is still in its MFMAs and epilogue;
//   * a wave owns 64 pixels x 64 channels (2 x 2 MFMA tiles): 24 ds_read_b128 per 24 MFMAs.
// Accumulation order = the strip kernel's (chunk outer, taps 0..8, k-steps), so the results are bit-identical to it.
// Optional second K segment (the block's 1x1 shortcut) with up to 32 channels (down1: the 8-channel image operand).
namespace ws {
constexpr int TM = 256;                                 // pixels per tile
constexpr int AROWS = 264;                              // strip rows per stage buffer: 258 used (pixel m0 - 1 .. m0 + 256)
constexpr int A_STAGE = AROWS * 64;                     // 16 896 B
constexpr int RS = 3;                                   // ring buffers (two stages in flight)
constexpr int W_BYTES = 9 * 2 * 64 * 64;                // resident weights [tap][chunk][cout][32 cin]
constexpr int W2_BYTES = 64 * 64;                       // second-segment weights [cout][<= 32 cin]
constexpr int EPI_BYTES = 4 * 32 * (64 + 4) * 4;        // gemm16_epilogue's wave tiles
constexpr int LDS_BYTES = W_BYTES + W2_BYTES + RS * A_STAGE + EPI_BYTES;
constexpr int AP = 5;                                   // DMA pieces per thread per stage (the fifth: wave 0 only)
}  // namespace ws

__global__ __launch_bounds__(256, 1) void conv64_ws_kernel(const IGemm p, int ntiles) {
  using namespace ws;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Wres = smem;
  unsigned char* W2 = smem + W_BYTES;
  unsigned char* ring = W2 + W2_BYTES;
  unsigned char* epi = ring + RS * A_STAGE;
  constexpr int ROWB = 64, CPR = 4, RSH = 2, KS = 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int lr = lane & 31, lh = lane >> 5;
  const int M = p.B * p.H * p.W, HW = p.H * p.W;
  // Tile walk: in round i the workgroups of the grid hold the G CONSECUTIVE tiles [i G, (i + 1) G) -- so at any moment the chip
  // streams one contiguous 4 MiB stretch of the tensor, spread over every memory channel (contiguous per-workgroup ranges put
  // all 256 CUs on addresses exactly 1 MiB apart, i.e. on the same channels: 556 us instead of 378 for the strip kernel) -- and
  // within a round each XCD (workgroup id mod 8) owns a contiguous eighth, so vertically adjacent tiles share that XCD's L2.
  const int G = gridDim.x;
  const int slot = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t0 = slot;                                  // tiles t0, t0 + G, t0 + 2 G, ... < ntiles
  if (t0 >= ntiles) return;
  const int my_tiles = (ntiles - 1 - t0) / G + 1;
  const int NS = 6 + (p.Cinb ? 1 : 0);                  // stages per tile
  const __bf16* A16 = reinterpret_cast<const __bf16*>(p.A16);
  const __bf16* W16 = reinterpret_cast<const __bf16*>(p.Wt16);
  const __bf16* A16b = reinterpret_cast<const __bf16*>(p.A16b);
  const __bf16* W16b = reinterpret_cast<const __bf16*>(p.Wt16b);
  const unsigned lds0 = (unsigned)(size_t)smem;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A16), 0, (int)((size_t)M * p.lda16 * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W16), 0, 9 * 64 * 64 * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t rsAb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Cinb ? A16b : A16), 0,
                                                                  p.Cinb ? (int)((size_t)M * p.lda16b * 2) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsWb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Cinb ? W16b : W16), 0, p.Cinb ? 64 * p.Cinb * 2 : 0, 0x00020000);

  // ---- resident weights: LDS piece q = ((tap * 2 + ck) * 64 + cout) * 4 + pos holds global chunk pos ^ key(cout)
#pragma unroll
  for (int i = 0; i < W_BYTES / 16 / 256; ++i) {
    const int q = i * 256 + tid, pos = q & 3, cout = (q >> 2) & 63, tc = q >> 8, ck = tc & 1, tap = tc >> 1;
    const int chunk = pos ^ ((cout >> RSH) & (CPR - 1));
    const unsigned v = (unsigned)(((tap * 64 + cout) * 64 + ck * 32 + 8 * chunk) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(size_t)(lds0 + (i * 256 + wave_s * 64) * 16), 16, v, 0, 0, 0);
  }
  if (p.Cinb) {   // [cout][Cinb <= 32] -> 64-byte rows, chunks past Cinb are zeros
    const int pos = tid & 3, cout = tid >> 2, chunk = pos ^ ((cout >> RSH) & (CPR - 1));
    const unsigned v = 8 * chunk < p.Cinb ? (unsigned)((cout * p.Cinb + 8 * chunk) * 2) : BUF_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWb, (lds_ptr_t)(size_t)(lds0 + W_BYTES + wave_s * 1024), 16, v, 0, 0, 0);
  }

  // ---- issue side: per-lane offsets of the tile whose stages are being fetched (recomputed when the issue side enters a tile)
  int a_chunk[AP];
#pragma unroll
  for (int j = 0; j < AP; ++j) a_chunk[j] = (tid & 3) ^ ((((j * 256 + tid) >> 2) >> RSH) & (CPR - 1));
  unsigned av0[AP], av1[AP], av2[AP], avb[AP];
  auto set_tile = [&](int tile) {
    const int m0 = tile * TM;
    const unsigned rowb = (unsigned)p.W * (unsigned)p.lda16 * 2u;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int row = (j * 256 + tid) >> 2, m = m0 - 1 + row;
      const bool ok = row < TM + 2 && m >= 0 && m < M;
      const int mm = ok ? m : 0;
      const int b = (int)p.fd_hw.div((unsigned)mm), y = (int)p.fd_w.div((unsigned)(mm - b * HW));
      const unsigned base = ((unsigned)mm * (unsigned)p.lda16 + 8u * a_chunk[j]) * 2u;
      av0[j] = (ok && y >= 1) ? base - rowb : BUF_OOB;
      av1[j] = ok ? base : BUF_OOB;
      av2[j] = (ok && y + 1 < p.H) ? base + rowb : BUF_OOB;
      avb[j] = (ok && 8 * a_chunk[j] < p.Cinb) ? ((unsigned)mm * (unsigned)p.lda16b + 8u * a_chunk[j]) * 2u : BUF_OOB;
    }
  };
  auto issue = [&](int stage, int buf) {                // stage 0..5: (chunk, kernel row); 6: the 1x1 second segment
    if (GEMM_DBG(p) & 1) return;                        // (experiments build: dissection run without the activation loads)
    const unsigned sbw = lds0 + W_BYTES + W2_BYTES + buf * A_STAGE + wave_s * 1024;
    if (stage >= 6) {
#pragma unroll
      for (int j = 0; j < AP; ++j) {
        if (j == AP - 1) {   // rows 256 .. 263: wave 0, lanes 0 .. 31 (the other half of the piece would land past the buffer)
          if (wave_s == 0 && lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsAb, (lds_ptr_t)(size_t)(sbw + j * 4096), 16, avb[j], 0, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsAb, (lds_ptr_t)(size_t)(sbw + j * 4096), 16, avb[j], 0, 0, 0);
        }
      }
      return;
    }
    const int ck = stage / 3, dyi = stage - ck * 3;
    const unsigned so = (unsigned)(ck * 64);
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const unsigned v = dyi == 0 ? av0[j] : (dyi == 1 ? av1[j] : av2[j]);
      if (j == AP - 1) {
        if (wave_s == 0 && lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(size_t)(sbw + j * 4096), 16, v, so, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(size_t)(sbw + j * 4096), 16, v, so, 0, 0);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][nt][r] = 0.f;
  const int b_key = (lr >> RSH) & (CPR - 1);
  bf16x8 zero8;
#pragma unroll
  for (int j = 0; j < 8; ++j) zero8[j] = (__bf16)0.f;

  // ---- prologue: the first two stages of the first tile are in flight before the loop
  const int total = my_tiles * NS;
  int itile = t0, istage = 0;                           // issue side (plain locals updated in line: a lambda capturing them by
  set_tile(itile);                                      //  reference put them -- and 30 scratch round trips per tile -- on the stack)
#define WS_ISSUE_NEXT(gg)                                                                                   \
  do {                                                                                                      \
    issue(istage, (gg) % RS);                                                                               \
    if (++istage == NS) { istage = 0; itile += G; if (itile < ntiles && !(GEMM_DBG(p) & 32)) set_tile(itile); } \
  } while (0)
  WS_ISSUE_NEXT(0);
  if (total > 1) WS_ISSUE_NEXT(1);

  int g = 0;
  for (int tile = t0; tile < ntiles; tile += G) {
    const int m0 = tile * TM;
    bool xl[2], xr[2], border[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int px = (m0 + wave * 64 + rb * 32 + lr) - (int)p.fd_w.div((unsigned)(m0 + wave * 64 + rb * 32 + lr)) * p.W;
      xl[rb] = px > 0; xr[rb] = px + 1 < p.W;
      border[rb] = __any(!xl[rb] || !xr[rb]) != 0;
    }
    for (int s = 0; s < NS; ++s, ++g) {
      // stage g has landed once at most the pieces of ONE younger stage of this wave are outstanding (wave 0 issues 5, the
      // others 4 per stage); the barrier extends that to every wave's pieces and retires all reads of the buffer refilled next
      if (!(GEMM_DBG(p) & 64)) {
      if (g + 1 < total) {
        if (wave_s == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      }
      if (g + 2 < total) WS_ISSUE_NEXT(g + 2);
      const unsigned char* st = ring + (g % RS) * A_STAGE;
      const bool seg2 = s >= 6;
      const int ck = s / 3, dyi = s - ck * 3;
      if (GEMM_DBG(p) & 4) continue;                    // (dissection run without fragment reads and MFMAs)
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        if (seg2 && dxi != 1) continue;
        const unsigned char* wt = seg2 ? W2 : Wres + ((dyi * 3 + dxi) * 2 + ck) * (64 * ROWB);
        bf16x8 af[2][KS], bfr[2][KS];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int row = wave * 64 + rb * 32 + lr + dxi, a_key = (row >> RSH) & (CPR - 1);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            af[rb][ks] = *reinterpret_cast<const bf16x8*>(st + row * ROWB + (((2 * ks + lh) ^ a_key) << 4));
            if (border[rb] && !seg2 && ((dxi == 0 && !xl[rb]) || (dxi == 2 && !xr[rb]))) af[rb][ks] = zero8;
          }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            bfr[nt][ks] = *reinterpret_cast<const bf16x8*>(wt + (nt * 32 + lr) * ROWB + (((2 * ks + lh) ^ b_key) << 4));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[rb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][ks], bfr[nt][ks], acc[rb][nt], 0, 0, 0);
      }
    }
    // ---- epilogue of the tile: the shared fused epilogue, once per 32-row block of the wave (it indexes rows as m0 + wave * 32)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      gemm16_epilogue<2>(p, acc[rb], reinterpret_cast<float*>(epi) + wave * epi_ctw<2>(), m0 + wave * 64 + rb * 32, 0, M, HW, 1, 0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][nt][r] = 0.f;
    }
  }
}

#undef WS_ISSUE_NEXT

bool conv64_ws_supported(const IGemm& p) {
  const int64_t M = (int64_t)p.B * p.H * p.W;
  const size_t lim = size_t(1) << 31;
  return p.taps == 9 && p.Cin == 64 && p.Cout == 64 && !p.f8 && !p.pk_q && !p.sel_C && p.lda16 >= 64 && (p.lda16 % 8) == 0 &&
         (p.Cinb == 0 || (p.Cinb <= 32 && (p.Cinb % 8) == 0 && p.A16b && p.Wt16b && (p.lda16b % 8) == 0)) &&
         (size_t)M * p.lda16 * 2 < lim && (!p.Cinb || (size_t)M * p.lda16b * 2 < lim) && M >= 2 * 256 * 256 &&
         exp_switch("DDPMIR_WS64");
}

void launch_conv64_ws(const IGemm& p, int64_t M, hipStream_t stream) {
  static PerDeviceOnce attr;
  static int n_cu = 0;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv64_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ws::LDS_BYTES));
    int dev = 0;
    hipDeviceProp_t prop;
    DDPMIR_HIP_CHECK(hipGetDevice(&dev));
    DDPMIR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount;
  });
  const int ntiles = (int)cdiv64(M, ws::TM);
  const int grid = std::min(ntiles, n_cu > 0 ? n_cu : 256);
  IGemm q = p;
  q.nmajor = 0; q.tap_pack = 0;
  hipLaunchKernelGGL(conv64_ws_kernel, dim3(grid), dim3(256), ws::LDS_BYTES, stream, q, ntiles);
}
#endif  // DDPMIR_EXPERIMENTS

template <int NT>
void launch_strip(const IGemm& p, int64_t M, hipStream_t stream) {
  constexpr int STAGE = (BM + 8) * 64 + 3 * 64 * 64, STAGE_B = (BM + 16) * 64 + 3 * 64 * 64;
  constexpr int HN = NT < 2 ? NT : 2;
  constexpr int EPI = 4 * 32 * (HN * 32 + 4) * 4;
  constexpr int LDS = 2 * STAGE > EPI ? 2 * STAGE : EPI, LDS_B = 2 * STAGE_B > EPI ? 2 * STAGE_B : EPI;
  static PerDeviceOnce attr;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_kernel<NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_kernel<NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
  });
  const int gx = (int)cdiv64(M, BM), gy = cdiv(p.Cout, NT * 32);
  IGemm q = p;
  q.nmajor = 0; q.tap_pack = 0;
  q.epi_direct = p.epi_direct_opt ? epilogue_direct_mode(p, NT * 32) : 0;   // (the strip kernel never splits K)
  // descriptor-addressed loads: whole 32-channel stages and operands below 2 GiB (32-bit offsets with headroom for soffset)
  const size_t lim = size_t(1) << 31;
  const bool buf = (p.Cin % 32) == 0 && (size_t)M * p.lda16 * 2 < lim && (size_t)9 * p.Cout * p.Cin * 2 < lim &&
                   (!p.Cinb || ((size_t)M * p.lda16b * 2 < lim && (p.Cinb % 8) == 0)) && !exp_switch("DDPMIR_NO_BUFLOAD");
  // three workgroups share a CU (LDS); with more tiles than that the workgroups walk them (gridDim.x a multiple of 8 like gx,
  // so a workgroup's tiles keep its XCD residue)
  static int n_cu = 0;
  static PerDeviceOnce cu_once;
  cu_once.run([] {
    int dev = 0;
    hipDeviceProp_t prop;
    DDPMIR_HIP_CHECK(hipGetDevice(&dev));
    DDPMIR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount;
  });
  const int resident = 3 * (n_cu > 0 ? n_cu : 256) / gy;
  const int grid_x = (gx > resident && !exp_switch("DDPMIR_NO_PERSIST")) ? std::max(8, resident & ~7) : gx;
  if (buf) hipLaunchKernelGGL((conv3x3_strip_kernel<NT, true>), dim3(grid_x, gy, 1), dim3(256), LDS_B, stream, q, gx);
  else hipLaunchKernelGGL((conv3x3_strip_kernel<NT, false>), dim3(grid_x, gy, 1), dim3(256), LDS, stream, q, gx);
}

// out[m][n] = epilogue(sum_s ws[s][m][n]) (s ascending): bias, time bias, activation, factors, residual; optional bf16 / e4m3 copy
__global__ void splitk_reduce_k(const IGemm p, int nsplit, int64_t M, int HW) {
  const int C4 = p.Cout / 4;
  const int64_t total = M * C4;
  __bf16* out16 = reinterpret_cast<__bf16*>(p.out16);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / C4;
    const int n = (int)(i - m * C4) * 4;
    float4 v = *reinterpret_cast<const float4*>(p.splitk_ws + (size_t)m * p.Cout + n);
    for (int sidx = 1; sidx < nsplit; ++sidx) {
      const float4 t = *reinterpret_cast<const float4*>(p.splitk_ws + ((size_t)sidx * M + m) * p.Cout + n);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (p.tb) {
      const float4 t = *reinterpret_cast<const float4*>(p.tb + (size_t)(m / HW) * p.ldtb + n);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    // (the rest of the main kernel's epilogue, in its order: activation, elementwise factor, per-sample factor)
    if (p.epi_act != ACT_NONE) { v.x = act16(v.x, p.epi_act); v.y = act16(v.y, p.epi_act); v.z = act16(v.z, p.epi_act); v.w = act16(v.w, p.epi_act); }
    if (p.mul) { const float4 t = *reinterpret_cast<const float4*>(p.mul + (size_t)m * p.ldmul + n); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
    if (p.rowscale) { const float rs = p.rowscale[m / HW]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
    if (p.res) { const float4 t = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    else if (p.res16) {
      const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.res16) + (size_t)m * p.ldres + n);
      v.x += (float)t[0]; v.y += (float)t[1]; v.z += (float)t[2]; v.w += (float)t[3];
    }
    if (p.out) *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldo + n) = v;
    if (out16) {
      bf16x4 o;
      o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(out16 + (size_t)m * p.ldo16 + n) = o;
    }
    if (p.out8) *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(p.out8) + (size_t)m * p.ldo8 + n) = pack_e4m3x4(v);
  }
}

// fp32 [M][ldx] -> bf16 [M][ld16] with the GroupNorm affine + activation of the consuming conv applied once
// (x*scale[b][c] + shift[b][c], then act); optionally also a raw bf16 copy (input of the 1x1 shortcut conv).
// Large tensors (every thread has work in every sample): a thread keeps its 8-channel slot, walks sample by sample with its
// scale / shift in registers (instead of four 16-byte loads per 16 bytes of payload) and has two pixels in flight.
__global__ void gn_apply16_k(const float* __restrict__ x, int ldx, int HW, int C, int Cs, const float* __restrict__ scale,
                             const float* __restrict__ shift, int act, __bf16* __restrict__ out, int ld16,
                             __bf16* __restrict__ raw, int ldraw, int64_t M, double* __restrict__ zero, int nzero,
                             unsigned char* __restrict__ out8, int ld8, const __bf16* __restrict__ x16in, FastDiv fd_c8, FastDiv fd_hw) {
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < nzero; i += blockDim.x) zero[i] = 0.0;
  const int C8 = (C + 7) / 8;
  const int64_t total = M * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // raw input values of 8 channels of row m
  auto load_raw = [&](int64_t m, int c, float r[8]) {
    if (x16in && c + 8 <= C) {                           // bf16 input: one 16-byte load per thread
      const bf16x8 tb = *reinterpret_cast<const bf16x8*>(x16in + m * ldx + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (float)tb[j];
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c + 4 * h < C) {                               // C is a multiple of 4
        if (x16in) {
          const bf16x4 tb = *reinterpret_cast<const bf16x4*>(x16in + m * ldx + c + 4 * h);
          t = make_float4((float)tb[0], (float)tb[1], (float)tb[2], (float)tb[3]);
        } else {
          t = *reinterpret_cast<const float4*>(x + m * ldx + c + 4 * h);
        }
      }
      r[4 * h] = t.x; r[4 * h + 1] = t.y; r[4 * h + 2] = t.z; r[4 * h + 3] = t.w;
    }
  };
  auto finish = [&](int64_t m, int c, const float r[8], const float4 s[2], const float4 f[2]) {
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool ok = c + 4 * h < C;
      v[4 * h] = ok ? act16(fmaf(r[4 * h], s[h].x, f[h].x), act) : 0.f;
      v[4 * h + 1] = ok ? act16(fmaf(r[4 * h + 1], s[h].y, f[h].y), act) : 0.f;
      v[4 * h + 2] = ok ? act16(fmaf(r[4 * h + 2], s[h].z, f[h].z), act) : 0.f;
      v[4 * h + 3] = ok ? act16(fmaf(r[4 * h + 3], s[h].w, f[h].w), act) : 0.f;
    }
    bf16x8 o, q;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = (__bf16)v[j]; q[j] = (__bf16)r[j]; }
    if (out) *reinterpret_cast<bf16x8*>(out + m * ld16 + c) = o;
    if (raw) *reinterpret_cast<bf16x8*>(raw + m * ldraw + c) = q;
    if (out8) {   // e4m3 operand of an fp8 conv (C % 8 == 0 there)
      uint2 w;
      w.x = pack_e4m3x4(make_float4(v[0], v[1], v[2], v[3]));
      w.y = pack_e4m3x4(make_float4(v[4], v[5], v[6], v[7]));
      *reinterpret_cast<uint2*>(out8 + m * ld8 + c) = w;
    }
  };
  float4 ks[2], kf[2];
  ks[0] = ks[1] = make_float4(1.f, 1.f, 1.f, 1.f); kf[0] = kf[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale && (C % 8) == 0 && (stride % C8) == 0 && stride / C8 <= HW) {
    const int pstep = (int)(stride / C8);
    const int p0 = (int)fd_c8.div((unsigned)gtid), c = ((int)gtid - p0 * C8) * 8;
    const int nb = (int)(M / HW);
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ks[h] = *reinterpret_cast<const float4*>(scale + (size_t)b * Cs + c + 4 * h);
        kf[h] = *reinterpret_cast<const float4*>(shift + (size_t)b * Cs + c + 4 * h);
      }
      const int64_t mb = (int64_t)b * HW;
      int pp = p0;
      for (; pp + pstep < HW; pp += 2 * pstep) {
        float r0[8], r1[8];
        load_raw(mb + pp, c, r0);
        load_raw(mb + pp + pstep, c, r1);
        finish(mb + pp, c, r0, ks, kf);
        finish(mb + pp + pstep, c, r1, ks, kf);
      }
      if (pp < HW) {
        float r0[8];
        load_raw(mb + pp, c, r0);
        finish(mb + pp, c, r0, ks, kf);
      }
    }
    return;
  }
  const bool small = total < (int64_t(1) << 31);        // 32-bit indices with exact magic-number division (the usual case)
  for (int64_t i = gtid; i < total; i += stride) {
    const int64_t m = small ? (int64_t)fd_c8.div((unsigned)i) : i / C8;
    const int c = (int)(i - m * C8) * 8;
    const int b = small ? (int)fd_hw.div((unsigned)m) : (int)(m / HW);
    if (scale) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (c + 4 * h < C) {
          ks[h] = *reinterpret_cast<const float4*>(scale + (size_t)b * Cs + c + 4 * h);
          kf[h] = *reinterpret_cast<const float4*>(shift + (size_t)b * Cs + c + 4 * h);
        }
    }
    float r[8];
    load_raw(m, c, r);
    finish(m, c, r, ks, kf);
  }
}

}  // namespace
void launch_splitk_reduce(const IGemm& p, int nsplit, int64_t M, hipStream_t stream);
namespace {

template <int NT, int BK, int STAGES, int F8 = 0>
void launch_nt(const IGemm& p, int64_t M, hipStream_t stream) {
  constexpr int CPR = BK / 8;
  constexpr int BROWS = NT * 32 * CPR < 256 ? 256 / CPR : NT * 32;
  constexpr int STAGE = (BM + BROWS) * BK * 2;
  constexpr int HN = NT < 2 ? NT : 2;
  constexpr int EPI = 4 * (NT == 4 ? 64 * 36 : 32 * (HN * 32 + 4)) * 4;
  constexpr int LDS = STAGES * STAGE > EPI ? STAGES * STAGE : EPI;
  static PerDeviceOnce attr;
  attr.run([] {
    DDPMIR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_kernel<NT, BK, STAGES, F8>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  });
  const int gx = (int)cdiv64(M, BM), gy = cdiv(p.Cout, NT * 32);
  const bool tap_pack = !F8 && p.taps == 9 && p.Cin == 8 && p.lda16 == 8 && !p.Cinb;
  const int nit = tap_pack ? cdiv(9, BK / 8) : p.taps * cdiv(p.Cin, F8 ? 2 * BK : BK) + cdiv(p.Cinb, BK);
  int nsplit = 1;
  // split-K needs an epilogue that splitk_reduce_k can replay on the summed tile: bias, time bias, activation, factors, residual
  const bool linear = p.epi_mask == MASK_NONE && (p.out || p.out16 || p.out8) && !p.pk_q && !p.sel_C && !p.gn_sums && p.splitk_ws;
  if (linear && gx * gy <= 256 && nit >= 8 * (64 / BK)) {
    nsplit = std::min(std::min(16, nit / (4 * (64 / BK))), cdiv(512, gx * gy));
    nsplit = std::min<int64_t>(nsplit, (int64_t)(p.splitk_ws_bytes / ((size_t)M * p.Cout * sizeof(float))));
    if (nsplit < 1) nsplit = 1;
  }
  IGemm q = p;
  q.tap_pack = tap_pack;
  q.epi_direct = (NT <= 2 && nsplit == 1 && p.epi_direct_opt) ? epilogue_direct_mode(p, NT * 32) : 0;
  q.nmajor = gx * gy < 256;                             // few M tiles: the weight stream is what to keep inside one L2
  const dim3 grid = q.nmajor ? dim3(gy, gx, nsplit) : dim3(gx, gy, nsplit);
  hipLaunchKernelGGL((gemm16_kernel<NT, BK, STAGES, F8>), grid, dim3(256), LDS, stream, q);
  if (nsplit > 1) launch_splitk_reduce(p, nsplit, M, stream);
}

template <int NT>
void launch_bk(const IGemm& p, int64_t M, hipStream_t stream) {
  // BK = 32 (four workgroups per CU) when there are enough output tiles to co-schedule; the deep levels
  // with few tiles go split-K, where the longer BK = 64 stage halves the barrier count (measured: DESIGN.md 4.2)
  const int64_t tiles = cdiv64(M, BM) * cdiv(p.Cout, NT * 32);
  // ... and so does a long K loop with enough work per barrier: 3x3 convs with Cout <= 64 or K >= 2048 measured 5-15 %
  // faster at BK = 64 / two workgroups per CU than at BK = 32 / four (tools/gemm_inventory.py), 1x1 and short-K the reverse
  const int ktot = p.taps * p.Cin + p.Cinb;
  const bool no_longk = exp_switch("DDPMIR_GEMM_NO_LONGK");   // A/B switch
  const bool longk = !no_longk && p.taps == 9 && (NT <= 2 || ktot >= 2048);
  const bool bk64 = (GEMM_DBG(p) & 32) ? true : (GEMM_DBG(p) & 64) ? false : (tiles < 256 || longk);
  if (p.f8) {   // fp8 operands: the 3x3 convs with >= 64 channels; a stage holds 2 * BK K elements
    if constexpr (NT >= 2) {
      if (bk64 && p.Cin >= 128) launch_nt<NT, 64, 2, 1>(p, M, stream);
      else launch_nt<NT, 32, 2, 1>(p, M, stream);
      return;
    }
    throw std::runtime_error("gemm16: fp8 operands need Cout > 32");
  }
  if constexpr (NT <= 2) {
    // full-resolution 3x3 convs with <= 64 output channels: one strip of pixel rows per kernel row serves its three taps
    if (p.taps == 9 && tiles >= 256 && p.Cin >= 32 && !(p.taps == 9 && p.Cin == 8) && !p.pk_q && !exp_switch("DDPMIR_NO_STRIP") && !(GEMM_DBG(p) & ~8)) {
#ifdef DDPMIR_EXPERIMENTS
      if constexpr (NT == 2) {
        if (conv64_ws_supported(p)) { launch_conv64_ws(p, M, stream); return; }   // (experiment: persistent, weights resident in LDS)
      }
#endif
      launch_strip<NT>(p, M, stream);
      return;
    }
  }
#ifdef DDPMIR_EXPERIMENTS
  if constexpr (NT == 4) {   // deeper DMA rings at BK = 32 (tools/ab_switches.py): 3 stages / 3 workgroups per CU, 4 stages / 2
    const int sel = exp_value("DDPMIR_GEMM_RING");   // 3 or 4: every 128-column launch; 13 / 14: only where BK = 64 is chosen today
    if (sel == 3 || (sel == 13 && bk64)) { launch_nt<NT, 32, 3>(p, M, stream); return; }
    if (sel == 4 || (sel == 14 && bk64)) { launch_nt<NT, 32, 4>(p, M, stream); return; }
  }
#endif
  if (bk64) launch_nt<NT, 64, 2>(p, M, stream);
  else launch_nt<NT, 32, 2>(p, M, stream);
}

}  // namespace

void launch_splitk_reduce(const IGemm& p, int nsplit, int64_t M, hipStream_t stream) {
  hipLaunchKernelGGL(splitk_reduce_k, dim3((unsigned)std::min<int64_t>(cdiv64(M * (p.Cout / 4), 256), 2048)), dim3(256), 0, stream,
                     p, nsplit, M, p.H * p.W);
}

bool gemm16_can_fuse_gn(const IGemm& p, int groups) {
  const int64_t M = (int64_t)p.B * p.H * p.W;
  const int NTc = p.Cout <= 32 ? 1 : (p.Cout <= 64 ? 2 : 4);
  const int64_t tiles = cdiv64(M, BM) * cdiv(p.Cout, NTc * 32);
  // (a group must lie inside one epilogue pass of one workgroup -- 64 columns, 32 for the one-column-tile kernel: channels per
  //  group a divisor of the pass width, Cout a multiple of it)
  const int cpg = groups > 0 ? p.Cout / groups : 0, pass = p.Cout <= 32 ? 32 : 64;
  return tiles >= 256 /* never split-K */ && ((p.H * p.W) % 32) == 0 && groups > 0 && (p.Cout % groups) == 0 &&
         (cpg % 4) == 0 && cpg <= pass && (pass % cpg) == 0 && (p.Cout % pass) == 0 && !p.pk_q && !p.sel_C && (p.out || p.out16 || p.out8);
}

bool gemm16_supported(const IGemm& p) {
  const int al = p.f8 ? 16 : 8;       // a 16-byte chunk holds 16 e4m3 or 8 bf16 K elements
  // (an output is required: a launch with every output pointer NULL would run its epilogue into address 0)
  return p.A16 && p.Wt16 && p.zero16 && (p.out || p.out16 || p.out8 || p.pk_q) && (!p.out || p.ldo >= (p.sel_C ? p.sel_C : p.Cout)) &&
         (!p.out16 || p.ldo16 > 0) && (!p.out8 || p.ldo8 > 0) && ((!p.res && !p.res16) || p.ldres > 0) && (!p.mul || p.ldmul > 0) && (!p.tb || p.ldtb > 0) &&
         (p.Cin % al) == 0 && (p.lda16 % al) == 0 && (p.Cout % 4) == 0 &&
         p.store == STORE_NHWC && p.pro_mask == MASK_NONE && !p.scale && p.pro_act == ACT_NONE && !(p.f8 && p.pk_q);
}

void launch_gemm16(const IGemm& p, hipStream_t stream) {
  if (!gemm16_supported(p)) throw std::runtime_error("gemm16: unsupported operand layout");
  if (p.pk_q && !((p.pk_D == 8 || p.pk_D == 16 || p.pk_D == 32) && p.pk_heads > 0 && ((p.pk_heads * p.pk_D) % 64) == 0 &&
                  p.Cout == 3 * p.pk_heads * p.pk_D && p.taps == 1 && p.bias && p.pk_kv && p.pk_kmax && p.pk_qmax &&
                  ((p.H * p.W) % 32) == 0 && p.pk_nt * 32 == p.H * p.W && p.epi_act == ACT_NONE && !p.mul && !p.res && !p.res16 && !p.tb))
    throw std::runtime_error("gemm16: fused qkv packing needs head_dim 8, 16 or 32 with heads * head_dim a multiple of 64, whole 32-token tiles, plain epilogue");
  if (p.Cinb && !(p.A16b && p.Wt16b && (p.Cinb % 8) == 0 && (p.lda16b % 8) == 0))
    throw std::runtime_error("gemm16: second K segment needs bf16 operands with channel counts that are multiples of 8");
  if (p.gn_sums) {
    IGemm q = p; q.gn_sums = nullptr;
    if (!gemm16_can_fuse_gn(q, p.gn_groups)) throw std::runtime_error("gemm16: GroupNorm statistics cannot be fused for this shape");
  }
  if (p.sel_C && !(p.Cout == 2 * p.sel_C && (p.sel_C % 64) == 0))
    throw std::runtime_error("gemm16: gate selection needs Cout = 2 * sel_C with sel_C a multiple of 64");
  const int64_t M = (int64_t)p.B * p.H * p.W;
  if (M <= 0 || p.Cout <= 0) return;
  if (M >= (int64_t(1) << 31)) throw std::runtime_error("gemm16: more than 2^31 pixels in one launch");
  IGemm q = p;
  q.fd_hw = FastDiv((unsigned)(p.H * p.W));
  q.fd_w = FastDiv((unsigned)p.W);
  if (q.Cout <= 32) launch_bk<1>(q, M, stream);
  else if (q.Cout <= 64) launch_bk<2>(q, M, stream);
  else if (!launch_gemm256(q, stream)) launch_bk<4>(q, M, stream);   // Cout >= 256 3x3 convs: the eight-wave 256 x 256 tile where it applies
  DDPMIR_HIP_CHECK(hipGetLastError());
}

void launch_gn_apply16(const float* x, int ldx, int B, int HW, int C, const float* scale, const float* shift, int act,
                       void* out16, int ld16, void* raw16, int ldraw, hipStream_t s, double* zero, int nzero, void* out8, int ld8,
                       const void* x16in) {
  const int64_t M = (int64_t)B * HW;
  const int64_t total = M * ((C + 7) / 8);
  int64_t g = (total + 255) / 256;
  g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
  hipLaunchKernelGGL(gn_apply16_k, dim3((unsigned)g), dim3(256), 0, s, x, ldx, HW, C, round_up(C, 4), scale, shift, act,
                     reinterpret_cast<__bf16*>(out16), ld16, reinterpret_cast<__bf16*>(raw16), ldraw, M, zero, nzero,
                     reinterpret_cast<unsigned char*>(out8), ld8, reinterpret_cast<const __bf16*>(x16in), FastDiv((unsigned)((C + 7) / 8)),
                     FastDiv((unsigned)HW));
  DDPMIR_HIP_CHECK(hipGetLastError());
}

}  // namespace ddpmir

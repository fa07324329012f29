// Round-4 groundwork, NOT product code: a 256 x 256 x 64 bf16 tile with eight waves, LDS-DMA half-tiles that stay in flight
// across raw barriers (counted vmcnt, never 0 in the loop) and two wave groups staggered by one barrier, as a stand-alone
// program with its own reference check -- the structure DESIGN.md 9 (a) names as the next step for the 3x3 convs with
// Cout >= 256 (gemm16_kernel reaches 630-690 TFLOP/s there, the vendor GEMM 1.0-1.2 PFLOP/s on the plain equivalents:
// profiles/r02_gemm_vendor_yardstick.txt).  Two operand modes: plain  C[M][N] = A[M][K] . W[N][K]^T  and the implicit 3x3
// convolution over an NHWC tensor (K = 9 taps x Cin, channel chunk outer / tap inner, out-of-image taps from a zero page),
// each with and without the stagger.
//
//   build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form gemm256_stagger.hip -o gemm256_stagger
//   run:   timeout -k 5 60 ./gemm256_stagger            (prints TFLOP/s, max error against a CPU reference on sampled rows,
//                                                        and whether repeated launches are bit-identical = race screen)
//
// Geometry.  Workgroup = 512 threads = 8 waves as 2 (rows, wr) x 4 (columns, wc); a wave owns 128 x 64 of the tile = 4 x 2
// accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs).  A K tile (64 deep) is staged as FOUR half-tiles of 16 KiB, in the
// order  B0, A0, B1, A1 :  A-half h = the rows {wr*128 + h*64 .. +63} of both wave rows, B-half h = the columns
// {wc*64 + h*32 .. +31} of all four wave columns, so every wave finds its own sub-rows in every half.  The stream of half-tiles
// g = 4 t + kind lives in slot g mod 8 of a 128 KiB ring.  A K tile is four phases, quadrant order (A0,B0) (A0,B1) (A1,B1)
// (A1,B0); each phase reads ONE new register sub-tile from LDS (A: 8 ds_read_b128, B: 4; the B0 of the next tile is read in the
// fourth phase into the B register set the third phase freed -- so the body is unrolled over two K tiles and every register
// index is a compile-time constant), issues ONE half-tile (two DMA instructions per thread) five half-tiles ahead, and
// runs 8 MFMAs.
//
// Synchronisation.  Phase = [reads + DMA issue] s_barrier [MFMAs, counted wait] s_barrier.  Group 1 (wr = 1) executes one
// extra barrier before the loop (group 0 one after it), so while one group's MFMAs run the other group's LDS reads are served:
// 24 KiB of LDS reads per group-phase = 192 LDS cycles beside 256 matrix-pipe cycles.  In barrier intervals: group 0 reads
// phase ph in interval 2 ph, multiplies in 2 ph + 1; group 1 one interval later.
//   RAW: half-tile g is issued in the read section of phase g - 5.  `s_waitcnt vmcnt(4)` at the end of the MFMA section of
//   phase g - 3 (two younger half-tiles stay in flight) retires this wave's share; group 1 does so in interval 2 g - 4, so after
//   barrier 2 g - 4 every share has landed; the first read of g is in phase g - 1 (B0 of the next tile; the others in phase g or
//   later) = interval 2 g - 2 for group 0: one full barrier later (the "one barrier more" a staggered pair needs).  The
//   prologue issues g = 0..4 and waits vmcnt(4) + barrier (g = 0, 1, 2 landed) before B0 of tile 0 is read.
//   WAR: g overwrites the slot of g - 8, last read in phase <= g - 8 = (g - 5) - 3: three phases before the issue.
//   Tail: when fewer than two younger half-tiles exist the wait is vmcnt(2) / vmcnt(0).
// Every wave executes 8 KT + 2 barriers; there is no spin, flag or data-dependent loop.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Prm {
  const __bf16* A;      // plain: [M][K]; conv: NHWC [B][H][W][Cin]
  const __bf16* W;      // plain: [N][K]; conv: [9][N][Cin]
  __bf16* C;            // [M][N]
  float* Cws;           // split-K (gridDim.y > 1): fp32 partials [split][M][N] instead of C
  const unsigned char* zero;   // 64 KiB of zeros (conv: out-of-image taps)
  int M, N, K;          // K = 9 * Cin in conv mode (elements: bf16, or e4m3 bytes with F8)
  int H, Wd, Cin;
  unsigned long long* clk;   // [0] shader-clock cycles, [1] 100 MHz wall-clock ticks of workgroup 0's run: the effective clock of the launch
};

// s_memtime / s_memrealtime land in SGPRs: no vector register is held across the kernel
#define CLK_BEGIN const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_w0 = wall_clock64();
#define CLK_END(p) if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (p).clk) { (p).clk[0] = __builtin_readcyclecounter() - clk_c0; (p).clk[1] = wall_clock64() - clk_w0; }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int HT_BYTES = 16384;   // one half-tile: 128 rows x 128 bytes

// F8: operands are OCP e4m3 bytes.  Same LDS geometry (rows of 128 bytes = 128 K elements per tile); a lane's 32-byte
// fragment of v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales) is the chunk pair 4 ks + 2 lh, + 1: twice the K per MFMA cycle.
// D = how many half-tiles ahead of the phase that issues it a half-tile lies in the stream (5 or 7).  Every half-tile g is read
// in phase g - 1 exactly, so its slot is free from phase g on; g + 8 reuses it and is issued in phase g + 8 - D.  D = 7 issues it
// two phases after the slot's last read (group 1 read it in interval 2 g - 1 and waited for those reads at the start of interval
// 2 g; group 0 issues in interval 2 g + 2): the deepest legal ring -- D = 8 would issue in the interval in which group 1 is still
// waiting for its reads.  The wait at the end of phase ph must retire every half-tile <= ph + 3 (see RAW above), so D - 3
// half-tiles stay in flight across it: vmcnt(2 (D - 3)).
// DBG (timing only, results meaningless): 1 no stores, 2 no DMA inside the loop, 4 no fragment reads inside the loop, 8 no MFMAs.
// EPI: the bf16 tile goes through a wave-private 16 KiB LDS tile and leaves in 16-byte row-contiguous stores (the direct form
// stores 2 bytes per lane: 9 us of a 46 us e4m3 launch).
template <bool CONV, bool STAG, bool F8, int D, int DBG, bool EPI>
__global__ __launch_bounds__(512, 1) void gemm256(const Prm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware, bijective remap: the workgroups of one XCD get consecutive tiles
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int ntn = p.N >> 8;
  const int tm = wg / ntn, tn = wg - tm * ntn;
  const int m0 = tm << 8, n0 = tn << 8;
  constexpr int EB = F8 ? 1 : 2;                         // bytes per K element; a K tile is 128 bytes of every row
  const int KTall = (p.K * EB) >> 7;
  const int t0 = (int)((long long)KTall * blockIdx.y / gridDim.y);
  const int KT = (int)((long long)KTall * (blockIdx.y + 1) / gridDim.y) - t0, G = 4 * KT;   // this split's K tiles (host: even)
  const unsigned lds0 = (unsigned)(size_t)smem;

  // ---- per-thread DMA sources.  Slot s = j*512 + tid of a half-tile: LDS row i = s >> 3, 16-byte position s & 7 holds the
  // row's chunk (s & 7) ^ key(i), key(i) = (i >> 1) & 7 (rows of 128 bytes: two rows per 256-byte bank row)
  const unsigned char* srcA[2][2];
  const unsigned char* srcB[2][2];
  int ay[2][2], ax[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int s = j * 512 + tid, i = s >> 3, chunk = (s & 7) ^ ((i >> 1) & 7);
      const int row = m0 + (i >> 6) * 128 + h * 64 + (i & 63);
      const int col = n0 + (i >> 5) * 64 + h * 32 + (i & 31);
      const size_t rowb = (size_t)(CONV ? p.Cin : p.K) * EB;          // bytes per operand row
      ay[h][j] = ax[h][j] = 0;
      if (CONV) {
        const int hw = p.H * p.Wd, b = row / hw, rem = row - b * hw;
        ay[h][j] = rem / p.Wd; ax[h][j] = rem - ay[h][j] * p.Wd;
      }
      srcA[h][j] = reinterpret_cast<const unsigned char*>(p.A) + (size_t)row * rowb + chunk * 16;
      srcB[h][j] = reinterpret_cast<const unsigned char*>(p.W) + (size_t)col * rowb + chunk * 16;
    }
  const unsigned char* zero = p.zero + (((blockIdx.x & 7) * 512 + tid) << 4);

  // kind 0 = B0, 1 = A0, 2 = B1, 3 = A1 of K tile `t` into ring slot `slot`
  auto issue = [&](auto kind_c, int tl, int slot) {
    constexpr int KIND = decltype(kind_c)::value, h = KIND >> 1;
    constexpr bool isA = KIND & 1;
    const int t = t0 + tl;                               // K tile of the whole contraction
    size_t koff = (size_t)t * 128;                       // plain: 128 bytes of every row per tile
    int dy = 0, dx = 0;
    if (CONV) {
      const int ck = t / 9, tap = t - ck * 9;
      dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
      koff = isA ? (size_t)ck * 128 : (size_t)tap * p.N * p.Cin * EB + (size_t)ck * 128;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const void* g;
      if (isA) {
        if (CONV) {
          const bool ok = (unsigned)(ay[h][j] + dy) < (unsigned)p.H && (unsigned)(ax[h][j] + dx) < (unsigned)p.Wd;
          g = ok ? (const void*)(srcA[h][j] + koff + (ptrdiff_t)(dy * p.Wd + dx) * p.Cin * EB) : (const void*)zero;
        } else g = srcA[h][j] + koff;
      } else g = srcB[h][j] + koff;
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(lds0 + slot * HT_BYTES + (j * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };

  // ---- fragment reads: lane (lr, lh) reads row lr of a 32-row sub-tile, K chunk (2 ks + lh) ^ key, key = (lr >> 1) & 7
  const int v = (lh ^ (lr >> 1)) & 7;
  const unsigned char* fa = smem + (wr * 64 + lr) * 128;
  const unsigned char* fb = smem + (wc * 32 + lr) * 128;
  int coff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    coff[ks] = F8 ? ((4 * (ks >> 1) + 2 * lh + (ks & 1)) ^ ((lr >> 1) & 7)) << 4     // a[..][2 k8], a[..][2 k8 + 1] = one 32-byte fragment
                  : (v ^ (2 * ks)) << 4;

  bf16x8 a[2][4], b[2][4];
  if (DBG & 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[i][k][e] = (__bf16)0.f; b[i][k][e] = (__bf16)0.f; }
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto rdA = [&](int slot) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[mt][ks] = *reinterpret_cast<const bf16x8*>(fa + slot * HT_BYTES + mt * 4096 + coff[ks]);
  };
  auto rdB = [&](int slot, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(fb + slot * HT_BYTES + coff[ks]);
  };

  auto bar = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // phase Q8 = 4 * (tile parity) + quadrant of the pair of K tiles (2 u, 2 u + 1)
  auto phase = [&](auto q_c, int u) {
    constexpr int Q8 = decltype(q_c)::value, PAR = Q8 >> 2, q = Q8 & 3, X = PAR, Y = PAR ^ 1;
    const int t = 2 * u + PAR, ph = 4 * t + q;
    // ---------------- read section
    if constexpr (!(DBG & 4)) {
      if constexpr (q == 0) rdA(4 * PAR + 1);
      if constexpr (q == 1) rdB(4 * PAR + 2, b[Y]);
      if constexpr (q == 2) rdA(4 * PAR + 3);
      if constexpr (q == 3) { if (t + 1 < KT) rdB(4 * (PAR ^ 1), b[Y]); }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 2) && ph + D < G) issue(std::integral_constant<int, (q + D) & 3>{}, t + (q + D) / 4, (Q8 + D) & 7);
    bar();
    // ---------------- multiply section
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
    constexpr int MI = (q >= 2) ? 2 : 0, NI = (q == 1 || q == 2) ? 1 : 0, BS = (q == 0 || q == 3) ? X : Y;
    if constexpr (DBG & 8) {
    } else if constexpr (F8) {
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
    if (D == 7 && ph + 7 < G) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (D == 7 && ph + 6 < G) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (ph + 5 < G) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
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
  if constexpr (D == 7) {
    issue(std::integral_constant<int, 1>{}, 1, 5);
    issue(std::integral_constant<int, 2>{}, 1, 6);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // half-tiles 0, 1, 2 landed; 3 .. 6 in flight
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // half-tiles 0, 1, 2 landed; 3, 4 in flight
  }
  bar();
  rdB(0, b[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slot 0 is refilled (half-tile 8) before group 1's first multiply section
  if (STAG && wr == 1) bar();
  for (int u = 0; u < (KT >> 1); ++u) {
    phase(std::integral_constant<int, 0>{}, u);
    phase(std::integral_constant<int, 1>{}, u);
    phase(std::integral_constant<int, 2>{}, u);
    phase(std::integral_constant<int, 3>{}, u);
    phase(std::integral_constant<int, 4>{}, u);
    phase(std::integral_constant<int, 5>{}, u);
    phase(std::integral_constant<int, 6>{}, u);
    phase(std::integral_constant<int, 7>{}, u);
  }
  if (STAG && wr == 0) bar();
  CLK_END(p)
  if ((DBG & 1) && acc[0][0][0] != 12345.f) return;      // timing without the epilogue (the accumulators stay live)
  if (EPI && !p.Cws) {
    // every DMA has landed and every wave is past its last fragment read (vmcnt(0) + barrier of the last phase, group 0's extra
    // barrier): wave w takes ring bytes [16 KiB w, +16 KiB) as its [128 rows][64 columns] bf16 tile
    __bf16* ct = reinterpret_cast<__bf16*>(smem + wave * HT_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + j * 32 + lr] = (__bf16)acc[i][j][r];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const u32x4 vv = *reinterpret_cast<const u32x4*>(ct + row * 64 + ch * 8);
      *reinterpret_cast<u32x4*>(p.C + (size_t)(m0 + wr * 128 + row) * p.N + n0 + wc * 64 + ch * 8) = vv;
    }
    return;
  }

  // ---------------- epilogue: acc[ah*2+mt][bh][r] = row wr*128 + ah*64 + mt*32 + (r&3) + 8 (r>>2) + 4 lh, column wc*64 + bh*32 + lr
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = n0 + wc * 64 + j * 32 + lr;
        if (p.Cws) p.Cws[((size_t)blockIdx.y * p.M + row) * p.N + col] = acc[i][j][r];
        else p.C[(size_t)row * p.N + col] = (__bf16)acc[i][j][r];
      }
}

// ------------------------------------------------------------------------------------------------- 256 x 128 tile (Cout = 128)
// The same eight waves (2 x 4) on a 256 x 128 tile: a wave owns 128 x 32 (four accumulators).  A K tile is THREE half-tiles of
// 16 KiB -- B (all 128 columns: wave column wc reads rows wc*32 .. +31), A0, A1 -- and TWO phases of 8 MFMAs: (A0, B), (A1, B).
// Reads: phase p0(t) A0(t); phase p1(t) A1(t) and B(t+1).  With r(g) = the phase that reads half-tile g
// (B(t): 2t-1, A0(t): 2t, A1(t): 2t+1) the two rules of the staggered pair are (file header, in barrier intervals)
//   R1  g must be retired by the counted wait at the END of phase r(g) - 2 (group 1 executes that wait in interval 2 r(g) - 2,
//       group 0 first reads in interval 2 r(g));
//   R2  g may be issued from phase r(g') + 2 on, g' = the previous occupant of its slot.
// A ring of 6 slots (two tiles) would force issue and retirement of B(t) into the same phase; with NINE slots (three tiles,
// 144 KiB) p0(u) issues A0(u+2) and p1(u) issues A1(u+2), B(u+3) -- R2 with equality for all three -- and both waits leave
// exactly three half-tiles in flight: end of p0(u) keeps A1(u+1), B(u+2), A0(u+2); end of p1(u) keeps A0(u+2), A1(u+2), B(u+3):
// vmcnt(6).  Once a phase has nothing left to issue the wait is vmcnt(0) (the last three tiles).  The prologue issues
// g = 0 .. 6 (through B(2)), waits vmcnt(6) (B(0), A0(0), A1(0), B(1) landed) + barrier, and reads B(0).  The body is unrolled
// over three tiles so every slot index is a constant; B(t+1) is read into a second register set and becomes B at the end of p1.
template <bool CONV, bool STAG, bool F8>
__global__ __launch_bounds__(512, 1) void gemm256x128(const Prm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int ntn = p.N >> 7;
  const int tm = wg / ntn, tn = wg - tm * ntn;
  const int m0 = tm << 8, n0 = tn << 7;
  constexpr int EB = F8 ? 1 : 2;
  const int KT = (p.K * EB) >> 7;                        // host: a multiple of 3, >= 3
  const unsigned lds0 = (unsigned)(size_t)smem;

  const unsigned char* srcA[2][2];
  const unsigned char* srcB[2];
  int ay[2][2], ax[2][2];
  const size_t rowb = (size_t)(CONV ? p.Cin : p.K) * EB;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = j * 512 + tid, i = s >> 3, chunk = (s & 7) ^ ((i >> 1) & 7);
    srcB[j] = reinterpret_cast<const unsigned char*>(p.W) + (size_t)(n0 + i) * rowb + chunk * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = m0 + (i >> 6) * 128 + h * 64 + (i & 63);
      ay[h][j] = ax[h][j] = 0;
      if (CONV) {
        const int hw = p.H * p.Wd, b = row / hw, rem = row - b * hw;
        ay[h][j] = rem / p.Wd; ax[h][j] = rem - ay[h][j] * p.Wd;
      }
      srcA[h][j] = reinterpret_cast<const unsigned char*>(p.A) + (size_t)row * rowb + chunk * 16;
    }
  }
  const unsigned char* zero = p.zero + (((blockIdx.x & 7) * 512 + tid) << 4);

  // kind 0 = B, 1 = A0, 2 = A1 of K tile t into ring slot `slot`
  auto issue = [&](auto kind_c, int t, int slot) {
    constexpr int KIND = decltype(kind_c)::value, h = KIND == 2 ? 1 : 0;
    constexpr bool isA = KIND != 0;
    size_t koff = (size_t)t * 128;
    int dy = 0, dx = 0;
    if (CONV) {
      const int ck = t / 9, tap = t - ck * 9;
      dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
      koff = isA ? (size_t)ck * 128 : (size_t)tap * p.N * p.Cin * EB + (size_t)ck * 128;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const void* g;
      if (isA) {
        if (CONV) {
          const bool ok = (unsigned)(ay[h][j] + dy) < (unsigned)p.H && (unsigned)(ax[h][j] + dx) < (unsigned)p.Wd;
          g = ok ? (const void*)(srcA[h][j] + koff + (ptrdiff_t)(dy * p.Wd + dx) * p.Cin * EB) : (const void*)zero;
        } else g = srcA[h][j] + koff;
      } else g = srcB[j] + koff;
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(size_t)(lds0 + slot * HT_BYTES + (j * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };

  const int v = (lh ^ (lr >> 1)) & 7;
  const unsigned char* fa = smem + (wr * 64 + lr) * 128;
  const unsigned char* fb = smem + (wc * 32 + lr) * 128;
  int coff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    coff[ks] = F8 ? ((4 * (ks >> 1) + 2 * lh + (ks & 1)) ^ ((lr >> 1) & 7)) << 4 : (v ^ (2 * ks)) << 4;

  bf16x8 a[2][4], b[4], bn[4];
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  auto rdA = [&](int slot) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[mt][ks] = *reinterpret_cast<const bf16x8*>(fa + slot * HT_BYTES + mt * 4096 + coff[ks]);
  };
  auto rdB = [&](int slot, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(fb + slot * HT_BYTES + coff[ks]);
  };
  auto bar = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // phase Q6 = 2 * (tile within the triple) + (0: A0 x B, 1: A1 x B); t = 3 w + Q6 / 2; half-tile g = 3 t + kind sits in slot g mod 9
  auto phase = [&](auto q_c, int w) {
    constexpr int Q6 = decltype(q_c)::value, TI = Q6 >> 1, q = Q6 & 1;
    const int t = 3 * w + TI;
    if constexpr (q == 0) {
      rdA((3 * TI + 1) % 9);                                           // A0(t)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < KT) issue(std::integral_constant<int, 1>{}, t + 2, (3 * (TI + 2) + 1) % 9);
    } else {
      rdA((3 * TI + 2) % 9);                                           // A1(t)
      if (t + 1 < KT) rdB((3 * (TI + 1)) % 9, bn);                      // B(t+1)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < KT) issue(std::integral_constant<int, 2>{}, t + 2, (3 * (TI + 2) + 2) % 9);
      if (t + 3 < KT) issue(std::integral_constant<int, 0>{}, t + 3, (3 * (TI + 3)) % 9);
    }
    bar();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      typedef int i32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int k8 = 0; k8 < 2; ++k8) {
        const i32x8 b8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, b[2 * k8]), __builtin_bit_cast(i32x4, b[2 * k8 + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const i32x8 a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, a[mt][2 * k8]), __builtin_bit_cast(i32x4, a[mt][2 * k8 + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
          acc[2 * q + mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[2 * q + mt], 0, 0, 0, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[2 * q + mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], b[ks], acc[2 * q + mt], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (q == 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b[ks] = bn[ks];
    }
    // steady state: three half-tiles stay in flight; once this phase had nothing (or not everything) to issue, drain
    if (q == 0 ? (t + 2 < KT) : (t + 3 < KT)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
  };

  issue(std::integral_constant<int, 0>{}, 0, 0);
  issue(std::integral_constant<int, 1>{}, 0, 1);
  issue(std::integral_constant<int, 2>{}, 0, 2);
  issue(std::integral_constant<int, 0>{}, 1, 3);
  issue(std::integral_constant<int, 1>{}, 1, 4);
  issue(std::integral_constant<int, 2>{}, 1, 5);
  issue(std::integral_constant<int, 0>{}, 2, 6);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // B(0), A0(0), A1(0), B(1) landed
  bar();
  rdB(0, b);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bn[ks] = b[ks];
  if (STAG && wr == 1) bar();
  for (int w = 0; w < KT / 3; ++w) {
    phase(std::integral_constant<int, 0>{}, w);
    phase(std::integral_constant<int, 1>{}, w);
    phase(std::integral_constant<int, 2>{}, w);
    phase(std::integral_constant<int, 3>{}, w);
    phase(std::integral_constant<int, 4>{}, w);
    phase(std::integral_constant<int, 5>{}, w);
  }
  if (STAG && wr == 0) bar();
  CLK_END(p)
  // wave-private LDS tile [128 rows][32 columns] bf16 (8 KiB of the ring), then 16-byte row-contiguous stores
  __bf16* ct = reinterpret_cast<__bf16*>(smem + wave * HT_BYTES);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = (__bf16)acc[i][r];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 16 + (lane >> 2), ch = lane & 3;
    const u32x4 vv = *reinterpret_cast<const u32x4*>(ct + row * 32 + ch * 8);
    *reinterpret_cast<u32x4*>(p.C + (size_t)(m0 + wr * 128 + row) * p.N + n0 + wc * 32 + ch * 8) = vv;
  }
}

// the strip kernel lives in gemm256_strip.hip (built without -amdgpu-mfma-vgpr-form: its accumulators go to AGPRs), which includes this file
template <bool STAG, bool F8>
__global__ void gemm256_strip(const Prm p);

// ------------------------------------------------------------------------------------------------- host
static inline unsigned short f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline float e4m3f(unsigned char b) {   // OCP e4m3fn
  const int e = (b >> 3) & 15, m = b & 7;
  const float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
  return (b & 0x80) ? -v : v;
}

struct Shape { const char* name; int B, H, Cin, M, N, K; };

template <bool CONV, bool STAG, bool F8, int D = 5, int DBG = 0, bool EPI = false, int BN = 256>
static int run(const Shape& s, int nsplit = 1) {
  const int dbg = DBG;
  const int M = s.M, N = s.N, K = s.K, EB = F8 ? 1 : 2;
  const int KT = K * EB / 128;
  constexpr int BNW = BN == 257 ? 256 : BN;              // BN = 257: the strip kernel (256 x 256 tile)
  if (BN == 257 && (!CONV || KT % 6 || nsplit != 1)) { printf("%s: strip kernel needs conv, K tiles %% 6 == 0\n", s.name); return 1; }
  if (M % 256 || N % BNW || (K * EB) % 128 || KT % nsplit || (BN == 256 ? (KT / nsplit) % 2 || KT / nsplit < 2 : KT % 3 || nsplit != 1) ||
      (CONV && (s.Cin * EB) % 128)) {
    printf("%s: shape not tileable (K tiles %d, splits %d)\n", s.name, KT, nsplit); return 1;
  }
  const size_t nA = CONV ? (size_t)M * s.Cin : (size_t)M * K, nW = (size_t)N * K, nC = (size_t)M * N;
  std::vector<float> fA(nA), fW(nW);                     // the operand VALUES (exactly representable), for the reference
  std::vector<unsigned char> hA(nA * EB), hW(nW * EB);
  unsigned long long st = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(M * 31 + N * 7 + K);
  auto rnd16 = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)((st >> 40) & 0xFFFF); };
  auto fill = [&](std::vector<float>& f, std::vector<unsigned char>& h) {
    for (size_t i = 0; i < f.size(); ++i) {
      if (F8) {   // random e4m3 with exponent field <= 8 (|x| < 4): no NaN / overflow codes
        const unsigned r = rnd16();
        const unsigned char b = (unsigned char)(((r & 1) << 7) | ((((r >> 1) % 9) & 15) << 3) | ((r >> 8) & 7));
        h[i] = b; f[i] = e4m3f(b);
      } else {
        const unsigned short b = f2bf((float)rnd16() / 32768.f - 1.f);
        memcpy(&h[2 * i], &b, 2); f[i] = bf2f(b);
      }
    }
  };
  fill(fA, hA); fill(fW, hW);
  __bf16 *dA, *dW, *dC; unsigned char* dZ; float* dWs = nullptr;
  // the 64 KiB zero page sits behind A in the same allocation (the strip kernel addresses it as a 32-bit offset from A)
  const size_t zoff = (hA.size() + 255) & ~(size_t)255;
  CHECK(hipMalloc(&dA, zoff + 65536)); CHECK(hipMalloc(&dW, hW.size())); CHECK(hipMalloc(&dC, nC * 2));
  dZ = reinterpret_cast<unsigned char*>(dA) + zoff;
  if (nsplit > 1) CHECK(hipMalloc(&dWs, nC * 4 * nsplit));
  CHECK(hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dW, hW.data(), hW.size(), hipMemcpyHostToDevice));
  CHECK(hipMemset(dZ, 0, 65536));
  void* dOut = nsplit > 1 ? (void*)dWs : (void*)dC;
  const size_t outBytes = nsplit > 1 ? nC * 4 * nsplit : nC * 2;
  CHECK(hipMemset(dOut, 0xFF, outBytes));
  unsigned long long* dClk; CHECK(hipMalloc(&dClk, 16)); CHECK(hipMemset(dClk, 0, 16));
  Prm p{dA, dW, dC, dWs, dZ, M, N, K, s.H, s.H, s.Cin, dClk};
  void (*kern)(Prm);
  if constexpr (BN == 256) kern = gemm256<CONV, STAG, F8, D, DBG, EPI>;
  else if constexpr (BN == 257) kern = gemm256_strip<STAG, F8>;
  else kern = gemm256x128<CONV, STAG, F8>;
  constexpr int LDS_BYTES = BN == 256 ? 8 * HT_BYTES : BN == 257 ? 4 * HT_BYTES + 2 * (32768 + 1024) + 1024 : 9 * HT_BYTES;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const dim3 grid((M / 256) * (N / BNW), nsplit), block(512);
  hipStream_t stream; CHECK(hipStreamCreate(&stream));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // first launch -> reference copy; then repeated launches must reproduce it bit for bit (race screen)
  hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, stream, p);
  CHECK(hipGetLastError());
  CHECK(hipStreamSynchronize(stream));
  std::vector<unsigned char> hC(outBytes), hC2(outBytes);
  CHECK(hipMemcpy(hC.data(), dOut, outBytes, hipMemcpyDeviceToHost));
  int unequal_runs = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemsetAsync(dOut, 0xFF, outBytes, stream));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, stream, p);
    CHECK(hipStreamSynchronize(stream));
    CHECK(hipMemcpy(hC2.data(), dOut, outBytes, hipMemcpyDeviceToHost));
    if (memcmp(hC.data(), hC2.data(), outBytes)) ++unequal_runs;
  }
  const int NREP = 20;
  CHECK(hipEventRecord(e0, stream));
  for (int i = 0; i < NREP; ++i) hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, stream, p);
  CHECK(hipEventRecord(e1, stream));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / NREP, tf = 2.0 * M * N * (double)K / (us * 1e-6) / 1e12;
  unsigned long long hClk[2] = {0, 0};
  CHECK(hipMemcpy(hClk, dClk, 16, hipMemcpyDeviceToHost));
  const double ghz = hClk[1] ? (double)hClk[0] / ((double)hClk[1] * 10.0) : 0.0;   // cycles per 10 ns tick -> GHz (workgroup 0, K loop only)
  auto got_at = [&](int row, int n) -> double {
    if (nsplit == 1) { unsigned short h; memcpy(&h, &hC[((size_t)row * N + n) * 2], 2); return bf2f(h); }
    double sum = 0;
    for (int sp = 0; sp < nsplit; ++sp) { float f; memcpy(&f, &hC[(((size_t)sp * M + row) * N + n) * 4], 4); sum += f; }
    return sum;
  };
  // CPU reference on sampled rows (all columns): first / last rows of tiles, wave and sub-tile boundaries, and a stride
  std::vector<int> rows;
  for (int r : {0, 1, 31, 32, 63, 64, 127, 128, 191, 192, 255, 256, 257, 511}) if (r < M) rows.push_back(r);
  for (int r = M - 300; r < M; r += 37) rows.push_back(r);
  for (int i = 1; i < 40; ++i) rows.push_back((int)(((long long)i * 2654435761ll) % M));
  double max_err = 0, max_ref = 0; long long bad = 0;
  std::vector<double> ref(N);
  if (dbg) rows.clear();
  for (int row : rows) {
    for (int n = 0; n < N; ++n) ref[n] = 0;
    if (!CONV) {
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        const float *ar = &fA[(size_t)row * K], *wv = &fW[(size_t)n * K];
        for (int k = 0; k < K; ++k) acc += (double)ar[k] * wv[k];
        ref[n] = acc;
      }
    } else {
      const int hw = s.H * s.H, bi = row / hw, rem = row % hw, y = rem / s.H, x = rem % s.H;
      for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy < 0 || yy >= s.H || xx < 0 || xx >= s.H) continue;
        const float* ar = &fA[((size_t)(bi * s.H + yy) * s.H + xx) * s.Cin];
        for (int n = 0; n < N; ++n) {
          const float* wv = &fW[((size_t)tap * N + n) * s.Cin];
          double acc = 0;
          for (int c = 0; c < s.Cin; ++c) acc += (double)ar[c] * wv[c];
          ref[n] += acc;
        }
      }
    }
    for (int n = 0; n < N; ++n) {
      const double got = got_at(row, n), e = fabs(got - ref[n]);
      if (e > max_err) max_err = e;
      if (fabs(ref[n]) > max_ref) max_ref = fabs(ref[n]);
      if (!(e <= 0.008 * fabs(ref[n]) + 0.02)) ++bad;
    }
  }
  char tag[64];
  snprintf(tag, sizeof tag, "%s%s%s%s%s%s", BN == 128 ? " TILE-256x128" : BN == 257 ? " A-STRIPS" : "", EPI ? " lds-epilogue" : "", (DBG & 1) ? " no-stores" : "", (DBG & 2) ? " no-dma" : "", (DBG & 4) ? " no-ldsreads" : "",
           (DBG & 8) ? " no-mfma" : "");
  printf("%-46s %s %s %s D%d split %d%s %7.1f us %7.1f TF/s %.2f GHz | %zu rows: max err %.4f (max |ref| %.1f), bad %lld, differing reruns %d/3\n",
         s.name, CONV ? "conv " : "plain", F8 ? "e4m3" : "bf16", STAG ? "staggered" : "lockstep ", D, nsplit, tag, us, tf, ghz,
         rows.size(), max_err, max_ref, bad, unequal_runs);
  fflush(stdout);
  CHECK(hipFree(dA)); CHECK(hipFree(dW)); CHECK(hipFree(dC)); CHECK(hipFree(dClk));
  if (dWs) CHECK(hipFree(dWs));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1)); CHECK(hipStreamDestroy(stream));
  return (bad || (unequal_runs && !dbg)) ? 1 : 0;
}

#ifndef GEMM256_NO_MAIN
int main(int argc, char** argv) {
  const char* set = argc > 1 ? argv[1] : "bf16";
  // the 3x3 convs with Cout >= 256 of the headline (B=16, 256x256) and configs[4] (B=8, 512x512) forwards, and their plain twins
  const Shape c256{"B=16 64x64   256->256 (M 65536 K 2304 N 256)", 16, 64, 256, 16 * 64 * 64, 256, 2304};
  const Shape c512{"B=16 32x32   512->512 (M 16384 K 4608 N 512)", 16, 32, 512, 16 * 32 * 32, 512, 4608};
  const Shape d256{"B=8 128x128  256->256 (M 131072 K 2304 N 256)", 8, 128, 256, 8 * 128 * 128, 256, 2304};
  const Shape d512{"B=8  64x64   512->512 (M 32768 K 4608 N 512)", 8, 64, 512, 8 * 64 * 64, 512, 4608};
  const Shape e512{"B=16 16x16   512->512 (M 4096 K 4608 N 512)", 16, 16, 512, 16 * 16 * 16, 512, 4608};
  int fails = 0;
  if (!strcmp(set, "bf16")) {          // first data set (profiles/r03_gemm256_stagger_first_run.txt)
    for (const Shape* s : {&c256, &c512, &d256, &d512}) {
      fails += run<false, true, false>(*s);
      fails += run<false, false, false>(*s);
      fails += run<true, true, false>(*s);
      fails += run<true, false, false>(*s);
    }
  } else if (!strcmp(set, "fp8")) {    // e4m3 operands, and split-K where a shape has fewer than 256 tiles
    for (const Shape* s : {&c256, &c512, &d256, &d512}) fails += run<true, true, true>(*s);
    fails += run<false, true, true>(d512);
    fails += run<true, false, true>(d512);
    fails += run<true, true, false>(c512, 2);
    fails += run<true, true, true>(c512, 2);
    fails += run<true, true, false>(e512, 4);
    fails += run<true, true, false>(e512, 6);
    fails += run<true, true, true>(e512, 6);
  } else if (!strcmp(set, "depth")) {  // ring depth 5 vs 7 (half-tiles in flight across the waits: 2 vs 4), and the epilogue's share
    for (const Shape* s : {&c256, &d256, &d512}) {
      fails += run<true, true, false, 5>(*s);
      fails += run<true, true, false, 7>(*s);
      fails += run<true, true, true, 5>(*s);
      fails += run<true, true, true, 7>(*s);
    }
    fails += run<false, true, false, 7>(c256);
    fails += run<false, true, true, 7>(d512);
    fails += run<true, false, false, 7>(c256);
    fails += run<true, true, false, 7, 1>(c256);
    fails += run<true, true, true, 7, 1>(c256);
    fails += run<true, true, false, 7>(c512, 2);
    fails += run<true, true, true, 7>(e512, 6);
  } else if (!strcmp(set, "dissect")) {   // the LDS epilogue (checked), and where a launch's time goes (timing-only switches)
    fails += run<true, true, false, 5, 0, true>(c256);
    fails += run<true, true, true, 5, 0, true>(c256);
    fails += run<true, true, false, 5, 0, true>(d512);
    fails += run<true, true, true, 5, 0, true>(d512);
    fails += run<false, true, false, 5, 0, true>(c256);
    fails += run<true, true, false, 5, 1>(c256);
    fails += run<true, true, false, 5, 1 | 2>(c256);
    fails += run<true, true, false, 5, 1 | 4>(c256);
    fails += run<true, true, false, 5, 1 | 8>(c256);
    fails += run<true, true, false, 5, 1 | 2 | 4>(c256);
    fails += run<true, true, false, 5, 1 | 2 | 8>(c256);
    fails += run<true, true, false, 5, 1 | 4 | 8>(c256);
    fails += run<true, true, false, 5, 1 | 2 | 4 | 8>(c256);
    fails += run<true, false, false, 5, 1 | 2 | 4>(c256);
    fails += run<true, true, true, 5, 1>(c256);
    fails += run<true, true, true, 5, 1 | 2>(c256);
    fails += run<true, true, true, 5, 1 | 4>(c256);
    fails += run<true, true, true, 5, 1 | 2 | 4>(c256);
  } else if (!strcmp(set, "dma")) {   // the DMA stream alone: does a deeper ring raise its rate?  (timing only)
    fails += run<true, true, false, 5, 1>(c256);
    fails += run<true, true, false, 7, 1>(c256);
    fails += run<true, true, false, 5, 1 | 8>(c256);
    fails += run<true, true, false, 7, 1 | 8>(c256);
    fails += run<true, true, false, 5, 1 | 4 | 8>(c256);
    fails += run<true, true, false, 7, 1 | 4 | 8>(c256);
    fails += run<false, true, false, 5, 1 | 4 | 8>(c256);
    fails += run<false, true, false, 7, 1 | 4 | 8>(c256);
    fails += run<true, false, false, 7, 1 | 4 | 8>(c256);
    fails += run<true, true, true, 5, 1 | 4 | 8>(d512);
    fails += run<true, true, true, 7, 1 | 4 | 8>(d512);
    fails += run<true, true, true, 5, 1 | 2 | 4>(d512);
    fails += run<true, true, true, 7, 1>(d512);
  } else if (!strcmp(set, "n128")) {   // the 256 x 128 tile on the Cout = 128 convs (gemm16_kernel: 128.6 us / 257.9 us for the first two)
    const Shape f128{"B=16 128x128 128->128 (M 262144 K 1152 N 128)", 16, 128, 128, 16 * 128 * 128, 128, 1152};
    const Shape g128{"B=8 256x256  128->128 (M 524288 K 1152 N 128)", 8, 256, 128, 8 * 256 * 256, 128, 1152};
    const Shape h128{"B=16 128x128 256->128 (M 262144 K 2304 N 128)", 16, 128, 256, 16 * 128 * 128, 128, 2304};
    fails += run<true, true, false, 5, 0, false, 128>(f128);
    fails += run<true, false, false, 5, 0, false, 128>(f128);
    fails += run<true, true, true, 5, 0, false, 128>(f128);
    fails += run<true, true, false, 5, 0, false, 128>(h128);
    fails += run<true, true, true, 5, 0, false, 128>(h128);
    fails += run<true, true, false, 5, 0, false, 128>(c256);
    if (argc > 2) { fails += run<false, true, false, 5, 0, false, 128>(f128); fails += run<true, true, false, 5, 0, false, 128>(g128); }
  } else if (!strcmp(set, "n128b")) {  // 128 tiles of 256 x 256 = 256 tiles of 256 x 128: the K split avoided?
    fails += run<true, true, false, 5, 0, false, 128>(c512);
    fails += run<true, true, true, 5, 0, false, 128>(c512);
  } else { printf("usage: gemm256_stagger [bf16|fp8|depth|dissect|dma|n128|n128b]\n"); return 2; }
  printf(fails ? "FAILED: %d configurations\n" : "all configurations correct and reproducible\n", fails);
  return fails ? 1 : 0;
}
#endif

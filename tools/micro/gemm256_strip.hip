// The 256 x 256 x 64 eight-wave tile of gemm256_stagger.hip with its A operand staged as ROW STRIPS (one strip per channel chunk
// and kernel row serves the three taps of that row: 128 KiB staged per three K tiles instead of 192).  Includes gemm256_stagger.hip
// for the types, the plain-staging kernels (the A/B partner in this binary) and the host harness.
//
//   build: hipcc -O3 --offload-arch=gfx950 gemm256_strip.hip -o gemm256_strip     (NO -amdgpu-mfma-vgpr-form: with the 128
//          accumulator registers in VGPRs the kernel spills, and a scratch reload is a VMEM operation that would break the
//          counted vmcnt; the accumulators belong in AGPRs here)
//   run:   timeout -k 5 60 ./gemm256_strip
#define GEMM256_NO_MAIN
#include "gemm256_stagger.hip"

// ------------------------------------------------------------------------------------------------- 256 x 256 tile, A from row strips
// Written at the end of round 3 after the CPU checks (strip_index_check.py, schedule_check.py) and run once
// (profiles/r03_gemm256_strip_first_run.txt): correct and bit-reproducible, 82.2 us against 85.2 us for plain staging (B=16, 64x64,
// 256 -> 256, bf16).  DESIGN.md 9.2 derives it; in short:
//  * K order: 128-byte channel chunk ck outer, kernel row dy, then the three taps dx of that row (K tile T = 3 G + dx + 1,
//    group G = 3 ck + dy + 1).  Per group ONE strip is staged: LDS row rho = pixel m0 + rho of the tile shifted by dy image rows
//    (rho = 0 .. 255), row 256 = pixel m0 - 1, row 257 = pixel m0 + 256, row 258 = zeros; a pixel whose image row y has y + dy
//    outside the image comes from the zero page.  Tap dx reads row i + dx for output pixel i -- lane-addressed, so an output on
//    the left / right image border points at row 258 for dx = -1 / +1 and no fragment is masked.  Rows keep the swizzle key
//    (rho >> 1) & 7; a shifted run of 32 rows is still conflict-free.
//  * LDS: weight half-tiles B0 / B1 of tap T in slot (2 T + h) mod 4 of a 64 KiB ring, two strip buffers of 33 KiB, 1 KiB that
//    only ever receives dummy DMAs: 132 KiB.
//  * EVERY phase issues exactly two DMA instructions per thread, so "instructions issued after item X" is 2 x (phases since):
//    q1 of tap U: B0(U + 2); q3: B1(U + 2); q0 / q2 of the three taps of group Gc: strip Gc + 1 main rows (2 + 2 instructions),
//    its rows 256 .. 258 (lanes 0 .. 23 of every wave, all waves writing the same bytes; issued twice), then dummies (zero page
//    -> the dummy KiB) -- also wherever an item no longer exists (tail).  B items are read 6 phases after their issue, strip
//    pieces >= 8 after, and an item read in phase r must be retired by the wait at the end of phase r - 2 (rule R1): at the end
//    of an odd phase the newest item that must have landed was issued 4 phases ago, of an even phase 5: vmcnt(8) / vmcnt(10),
//    constant through prologue (which replays phases -12 .. -1 in order) and tail.  Slot reuse (R2: issue >= last read + 2):
//    B0(T) at 4 T - 7 after B0(T - 2) read at 4 T - 9; B1(T) at 4 T - 5 after 4 T - 7; strip Gc + 1 from phase 12 Gc after strip
//    Gc - 1 last read at 12 Gc - 2.  The body is unrolled over six taps (two groups): every slot index is a constant.
template <bool STAG, bool F8>
__global__ __launch_bounds__(512, 1) void gemm256_strip(const Prm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CLK_BEGIN
  constexpr int EB = F8 ? 1 : 2;
  constexpr int STRIP_BYTES = 32768 + 1024, STRIP0 = 4 * HT_BYTES, DUMMY = STRIP0 + 2 * STRIP_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int ntn = p.N >> 8;
  const int tm = wg / ntn, tn = wg - tm * ntn;
  const int m0 = tm << 8, n0 = tn << 8;
  const int KT = 9 * ((p.Cin * EB) >> 7), NG = KT / 3;   // host: KT a multiple of 6
  const int hw = p.H * p.Wd;
  const size_t rowb = (size_t)p.Cin * EB;
  const unsigned lds0 = (unsigned)(size_t)smem;
  // ---- DMA sources: ONE 32-bit lane offset per operand plus wave-uniform bases (a thread's four strip slots are 64 rows apart and
  // share chunk position and key, its two weight slots 128 columns apart), so no 64-bit address lives in a VGPR across the loop --
  // with 128 accumulator registers at two waves per SIMD (256 registers per lane in all) the loop has ~125 for everything else.
  // The zero page sits behind A in the same allocation (host), so "zeros" is also a 32-bit offset from the strip's base.
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A);
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
  const int rho0 = tid >> 3, chunk0 = (tid & 7) ^ ((rho0 >> 1) & 7);
  const unsigned offS = (unsigned)((size_t)(m0 + rho0) * rowb + chunk0 * 16);          // strip slot j: + j * 64 * rowb
  const unsigned zrel = (unsigned)(p.zero - Ab) + (unsigned)((((blockIdx.x & 7) * 512 + tid) << 4));
  // okm bit (2 j + d): the image row of pixel m0 + rho0 + 64 j has a neighbour row in direction d (0: above, 1: below)
  unsigned okm = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pix = m0 + rho0 + 64 * j, b = pix / hw, y = (pix - b * hw) / p.Wd;
    if (y > 0) okm |= 1u << (2 * j);
    if (y < p.H - 1) okm |= 1u << (2 * j + 1);
  }
  // rows 256 (pixel m0 - 1), 257 (pixel m0 + 256), 258 (zeros): lanes 0 .. 23 of every wave
  unsigned offX = 0, okx = 0;
  {
    const int ar = lane >> 3, rho = 256 + ar, chunk = (lane & 7) ^ ((rho >> 1) & 7);
    const int pix = ar == 0 ? m0 - 1 : m0 + 256;
    if (ar < 2 && pix >= 0 && pix < p.M) {
      const int b = pix / hw, y = (pix - b * hw) / p.Wd;
      offX = (unsigned)((size_t)pix * rowb + chunk * 16);
      okx = 4u | (y > 0 ? 1u : 0u) | (y < p.H - 1 ? 2u : 0u);
    }
  }
  // weight slot j of half h: row i = j*64 + (tid >> 3) of the half-tile = column n0 + (i >> 5)*64 + h*32 + (i & 31)
  const unsigned offB = (unsigned)((size_t)(n0 + (rho0 >> 5) * 64 + (rho0 & 31)) * rowb + chunk0 * 16);   // + (j*128 + h*32) * rowb

  auto dma = [&](const unsigned char* sbase, unsigned voff, unsigned lds_byte) {   // sbase wave-uniform
    __builtin_amdgcn_global_load_lds(sbase + voff, (lds_ptr_t)(size_t)(lds0 + lds_byte), 16, 0, 0);
  };
  auto issue_dummy = [&]() {   // two instructions that only keep the per-phase count (zero page -> the dummy KiB, which nobody reads)
    dma(Ab, zrel, DUMMY);
    dma(Ab, zrel, DUMMY);
  };
  // weight half-tile h of K tile T (exists iff T < KT) into ring slot (2 T + h) & 3
  auto issue_B = [&](int h_c, int T, int slot) {
    if (T >= KT) { issue_dummy(); return; }
    const int G = T / 3, dxi = T - 3 * G, ck = G / 3, dyi = G - 3 * ck;
    const unsigned char* wb = Wb + (size_t)(dyi * 3 + dxi) * p.N * rowb + (size_t)ck * 128 + (size_t)(h_c * 32) * rowb;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma(wb + (size_t)(j * 128) * rowb, offB, slot * HT_BYTES + (j * 512 + wave * 64) * 16);
  };
  // piece 0: main rows 0 .. 127, piece 1: rows 128 .. 255, piece 2: rows 256 .. 258 (twice) of the strip of group G into buffer `buf`
  auto issue_S = [&](int piece, int G, int buf) {
    if (G >= NG) { issue_dummy(); return; }
    const int ck = G / 3, dyi = G - 3 * ck, dy = dyi - 1;
    const ptrdiff_t soff = (ptrdiff_t)ck * 128 + (ptrdiff_t)dy * p.Wd * (ptrdiff_t)rowb;
    const unsigned sb = STRIP0 + buf * STRIP_BYTES;
    if (piece < 2) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * piece + jj;
        const ptrdiff_t boff = soff + (ptrdiff_t)(64 * j) * (ptrdiff_t)rowb;       // uniform
        const bool ok = dy == 0 || ((okm >> (2 * j + (dy > 0 ? 1 : 0))) & 1u);
        dma(Ab + boff, ok ? offS : zrel - (unsigned)boff, sb + (j * 512 + wave * 64) * 16);
      }
    } else {
      const bool ok = (okx & 4u) && (dy == 0 || ((okx >> (dy > 0 ? 1 : 0)) & 1u));
      const unsigned vo = ok ? offX : zrel - (unsigned)soff;
      if (lane < 24) { dma(Ab + soff, vo, sb + 32768); dma(Ab + soff, vo, sb + 32768); }
    }
  };

  // ---- fragment addresses: sub-tile (ah, mt) row i = wr*128 + ah*64 + mt*32 + lr; tap dxi reads strip row i + dxi - 1 (or 256 / 257 / 258)
  unsigned fbase[3][4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const int i = wr * 128 + sub * 32 + lr, pix = m0 + i, x = (pix % hw) % p.Wd;
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      int rho = i + dxi - 1;
      if (rho < 0) rho = 256;
      if (rho > 255) rho = 257;
      if ((dxi == 0 && x == 0) || (dxi == 2 && x == p.Wd - 1)) rho = 258;
      const int key = (rho >> 1) & 7;
      fbase[dxi][sub] = (unsigned)(rho * 128 + ((F8 ? ((2 * lh) ^ key) : (lh ^ key)) << 4));
    }
  }
  const unsigned fbB = (unsigned)((wc * 32 + lr) * 128 + ((F8 ? ((2 * lh) ^ ((lr >> 1) & 7)) : ((lh ^ (lr >> 1)) & 7)) << 4));

  bf16x8 a[2][4], b[2][4];
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto rdA = [&](int dxi, int ah, int buf) {
    const unsigned char* sbp = smem + STRIP0 + buf * STRIP_BYTES;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      unsigned fv = fbase[dxi][2 * ah + mt];
      asm volatile("" : "+v"(fv));            // keep the three XORs below inside the loop: hoisted, the 36 addresses spill
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const unsigned cx = F8 ? (unsigned)((4 * (ks >> 1) + (ks & 1)) << 4) : (unsigned)((2 * ks) << 4);
        a[mt][ks] = *reinterpret_cast<const bf16x8*>(sbp + (fv ^ cx));
      }
    }
  };
  auto rdB = [&](int slot, bf16x8 (&dst)[4]) {
    unsigned fv = fbB;
    asm volatile("" : "+v"(fv));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned cx = F8 ? (unsigned)((4 * (ks >> 1) + (ks & 1)) << 4) : (unsigned)((2 * ks) << 4);
      dst[ks] = *reinterpret_cast<const bf16x8*>(smem + slot * HT_BYTES + (fv ^ cx));
    }
  };
  auto bar = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // phase Q24 = 4 * (tap within the six of two groups) + quadrant; T = 6 u + C
  auto phase = [&](auto q_c, int u) {
    constexpr int Q24 = decltype(q_c)::value, C = Q24 >> 2, q = Q24 & 3;
    constexpr int GI = C / 3, dxi = C % 3, PAR = C & 1, X = PAR, Y = PAR ^ 1;   // group within the pair, tap, B register roles
    const int T = 6 * u + C, Gc = 2 * u + GI;
    // ---------------- read section
    if constexpr (q == 0) rdA(dxi, 0, GI);
    if constexpr (q == 1) rdB(2 * PAR + 1, b[Y]);                               // B1(T)
    if constexpr (q == 2) rdA(dxi, 1, GI);
    if constexpr (q == 3) { if (T + 1 < KT) rdB(2 * (PAR ^ 1), b[Y]); }          // B0(T+1)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (q == 1) issue_B(0, T + 2, 2 * PAR);                           // slot (2 (T+2) + 0) & 3 = 2 PAR
    if constexpr (q == 3) issue_B(1, T + 2, 2 * PAR + 1);
    if constexpr (q == 0 || q == 2) {
      constexpr int piece = 2 * dxi + (q >> 1);                                // 0 .. 5 over the group's six q0 / q2 phases
      if constexpr (piece < 3) issue_S(piece, Gc + 1, GI ^ 1); else issue_dummy();
    }
    bar();
    // ---------------- multiply section
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
    constexpr int MI = (q >= 2) ? 2 : 0, NI = (q == 1 || q == 2) ? 1 : 0, BS = (q == 0 || q == 3) ? X : Y;
    if constexpr (F8) {
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
    if constexpr (q & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    bar();
  };

  // ---------------- prologue = the issue lists of phases -12 .. -1 in order (two instructions each; see the header)
  issue_S(0, 0, 0);            // -12
  issue_dummy();               // -11
  issue_S(1, 0, 0);            // -10
  issue_dummy();               //  -9
  issue_S(2, 0, 0);            //  -8
  issue_B(0, 0, 0);            //  -7  B0(0)
  issue_dummy();               //  -6
  issue_B(1, 0, 1);            //  -5  B1(0)
  issue_dummy();               //  -4
  issue_B(0, 1, 2);            //  -3  B0(1)
  issue_dummy();               //  -2
  issue_B(1, 1, 3);            //  -1  B1(1)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // everything through B1(0) has landed (B0(1), B1(1) and two dummies in flight)
  bar();
  rdB(0, b[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (STAG && wr == 1) bar();
  for (int u = 0; u < KT / 6; ++u) {
    phase(std::integral_constant<int, 0>{}, u);  phase(std::integral_constant<int, 1>{}, u);  phase(std::integral_constant<int, 2>{}, u);
    phase(std::integral_constant<int, 3>{}, u);  phase(std::integral_constant<int, 4>{}, u);  phase(std::integral_constant<int, 5>{}, u);
    phase(std::integral_constant<int, 6>{}, u);  phase(std::integral_constant<int, 7>{}, u);  phase(std::integral_constant<int, 8>{}, u);
    phase(std::integral_constant<int, 9>{}, u);  phase(std::integral_constant<int, 10>{}, u); phase(std::integral_constant<int, 11>{}, u);
    phase(std::integral_constant<int, 12>{}, u); phase(std::integral_constant<int, 13>{}, u); phase(std::integral_constant<int, 14>{}, u);
    phase(std::integral_constant<int, 15>{}, u); phase(std::integral_constant<int, 16>{}, u); phase(std::integral_constant<int, 17>{}, u);
    phase(std::integral_constant<int, 18>{}, u); phase(std::integral_constant<int, 19>{}, u); phase(std::integral_constant<int, 20>{}, u);
    phase(std::integral_constant<int, 21>{}, u); phase(std::integral_constant<int, 22>{}, u); phase(std::integral_constant<int, 23>{}, u);
  }
  if (STAG && wr == 0) bar();
  CLK_END(p)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tail's dummies
  bar();
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
}


int main(int argc, char** argv) {
  const Shape c256{"B=16 64x64   256->256 (M 65536 K 2304 N 256)", 16, 64, 256, 16 * 64 * 64, 256, 2304};
  const Shape d256{"B=8 128x128  256->256 (M 131072 K 2304 N 256)", 8, 128, 256, 8 * 128 * 128, 256, 2304};
  const Shape d512{"B=8  64x64   512->512 (M 32768 K 4608 N 512)", 8, 64, 512, 8 * 64 * 64, 512, 4608};
  int fails = 0;
  if (argc > 1 && !strcmp(argv[1], "quick")) {               // one shape, strips against plain staging
    fails += run<true, true, false, 5, 0, false, 257>(c256);
    fails += run<true, true, false, 5, 0, true>(c256);
    fails += run<true, true, true, 5, 0, false, 257>(c256);
    printf(fails ? "FAILED: %d configurations\n" : "all configurations correct and reproducible\n", fails);
    return fails ? 1 : 0;
  }
  for (const Shape* s : {&c256, &d256, &d512}) {
    fails += run<true, true, false, 5, 0, true>(*s);          // plain staging, LDS epilogue (this build's flags)
    fails += run<true, true, false, 5, 0, false, 257>(*s);    // strips
    fails += run<true, true, true, 5, 0, true>(*s);
    fails += run<true, true, true, 5, 0, false, 257>(*s);
  }
  fails += run<true, false, false, 5, 0, false, 257>(c256);
  printf(fails ? "FAILED: %d configurations\n" : "all configurations correct and reproducible\n", fails);
  return fails ? 1 : 0;
}

// Which store pattern does a register-direct GEMM epilogue get away with?  A wave owns 32 pixels x 64 bf16 channels (128 B per
// pixel row, rows 128 B apart = an NHWC tensor with 64 channels) or 32 x 64 fp32 (256 B rows).  Patterns, all 16 bytes per lane:
//   full   : lane -> (row = it*8 + lane/8, 16-byte piece lane%8)         8 whole 128-byte rows per instruction (today's LDS-transposed epilogue)
//   tr     : lane -> (row = lane%32, piece 2*it + lane/32)                32 rows x 32 contiguous bytes per instruction (accumulators of the
//                                                                         swapped-operand MFMA after a v_permlane32_swap: no LDS round trip)
//   tr8    : lane -> (row = lane%32, 8 bytes at piece it, half lane/32)   32 rows x 16 contiguous bytes (the same without the swap)
//   col    : lane -> (column lane%32 (+32), rows 2 per instruction, 2 bytes per lane)   the un-transposed accumulator stored directly
// build: hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int ROWB>   // ROWB = bytes per pixel row (128: bf16 x 64, 256: fp32 x 64)
__global__ __launch_bounds__(256) void st(unsigned char* out, long long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long tile = blockIdx.x; tile * 128 < rows; tile += gridDim.x) {
    unsigned char* base = out + (tile * 128 + wave * 32) * ROWB;
    const u32x4 v = {(unsigned)tile, (unsigned)lane, 3u, 4u};
    if (MODE == 0) {
#pragma unroll
      for (int it = 0; it < 32 * ROWB / 1024; ++it) {
        const int idx = it * 64 + lane;
        *reinterpret_cast<u32x4*>(base + (idx / (ROWB / 16)) * ROWB + (idx % (ROWB / 16)) * 16) = v;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int it = 0; it < ROWB / 32; ++it)
        *reinterpret_cast<u32x4*>(base + (lane & 31) * ROWB + (2 * it + (lane >> 5)) * 16) = v;
    } else if (MODE == 2) {
#pragma unroll
      for (int it = 0; it < ROWB / 16; ++it)
        *reinterpret_cast<u32x2*>(base + (lane & 31) * ROWB + it * 16 + (lane >> 5) * 8) = u32x2{v[0], v[1]};
    } else if (ROWB == 128) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<unsigned short*>(base + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ROWB + (half * 32 + (lane & 31)) * 2) = (unsigned short)v[0];
    } else {   // fp32: 4 bytes per lane, 32 columns = 128 contiguous bytes per row
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<unsigned*>(base + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ROWB + (half * 32 + (lane & 31)) * 4) = v[0];
    }
  }
}

int main() {
  const long long rows = 8LL * 512 * 512;            // 64 -> 64 at 512 x 512, B = 8
  unsigned char* out; hipMalloc(&out, rows * 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[4] = {"full rows (LDS-transposed)", "tr (32 rows x 32 B)", "tr8 (32 rows x 16 B)", "col (2 / 4 B per lane)"};
  for (int rowb : {128, 256})
    for (int mode = 0; mode < 4; ++mode) {
      for (int grid : {768, 2048, 16384}) {
        auto launch = [&] {
          if (rowb == 128) { if (mode == 0) st<0, 128><<<grid, 256>>>(out, rows); else if (mode == 1) st<1, 128><<<grid, 256>>>(out, rows); else if (mode == 2) st<2, 128><<<grid, 256>>>(out, rows); else st<3, 128><<<grid, 256>>>(out, rows); }
          else { if (mode == 0) st<0, 256><<<grid, 256>>>(out, rows); else if (mode == 1) st<1, 256><<<grid, 256>>>(out, rows); else if (mode == 2) st<2, 256><<<grid, 256>>>(out, rows); else st<3, 256><<<grid, 256>>>(out, rows); }
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("row %3d B  %-30s grid %5d: %7.1f us  %6.2f TB/s\n", rowb, names[mode], grid, ms * 1e3, rows * rowb / ms / 1e9);
      }
    }
  return 0;
}

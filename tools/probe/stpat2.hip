// Experiment (not product): which property of k_gemm_nt_st's store stream costs bandwidth?  Same address order as stpat.hip PAT 0
// (wave = 4 slabs x a range of 16-row tiles, two 16-byte stores per lane and tile).
//   V0 global stores back to back        V1 buffer stores         V2 at most 6 stores in flight (s_waitcnt vmcnt(6) per tile)
//   V3 ~W VALU instructions between tiles   V4 = V2 + V3          V5 = V1 + V2 + V3
//   V6 W VALU instructions between the two stores of a tile    V7 = V6 with each store writing 8 whole rows (256 contiguous bytes
//   per quarter wave) instead of one half of 16 rows
//   hipcc --offload-arch=gfx950 -O3 tools/probe/stpat2.hip -o tools/probe/stpat2 && tools/probe/stpat2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int V, int W>
__global__ __launch_bounds__(256) void k(u32x4* __restrict__ y, long M, int nchunks, int tiles_per_item) {
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long item = (long)blockIdx.x * 4 + wave;
  const long mtiles = M / 16, nranges = (mtiles + tiles_per_item - 1) / tiles_per_item;
  if (item >= nranges * nchunks) return;
  const int chunk = item % nchunks; const long range = item / nchunks;
  const long ss16 = M * 2;   // slab stride in 16-byte units
  long mt = range * tiles_per_item, me = mt + tiles_per_item < mtiles ? mt + tiles_per_item : mtiles;
  u32x4* wbase = y + (long)(chunk * 4) * ss16 + mt * 32;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, 0x80000000, 0x00020000);
  const unsigned lane_off = (unsigned)((q * ss16 + j * 2) * 16);
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = lane * 0.01f + i;
  const int nt = (int)(me - mt);
  for (int t = 0; t < nt; ++t) {
    if (V == 3 || V == 4 || V == 5) {
#pragma unroll
      for (int w = 0; w < W / 8; ++w)
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = f[i] * 1.0001f + 0.5f;
    }
    u32x4 d = {(unsigned)t, __float_as_uint(f[0] + f[1] + f[2] + f[3]), __float_as_uint(f[4] + f[5] + f[6] + f[7]), 4u};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (V == 1 || V == 5) __builtin_amdgcn_raw_buffer_store_b128(d, rc, lane_off + (unsigned)t * 512u + s * 16u, 0, 0);
      else if (V == 7) wbase[(long)q * ss16 + (t * 16 + (j >> 1) + 8 * s) * 2 + (j & 1)] = d;   // 256 contiguous bytes per quarter wave
      else wbase[(long)q * ss16 + (t * 16 + j) * 2 + s] = d;
      if ((V == 6 || V == 7) && s == 0) {   // the two stores of a tile W VALU instructions apart
#pragma unroll
        for (int w = 0; w < W / 8; ++w)
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = f[i] * 1.0001f + 0.5f;
        d[1] = __float_as_uint(f[0] + f[1] + f[2] + f[3]); d[2] = __float_as_uint(f[4] + f[5] + f[6] + f[7]);
      }
    }
    if (V == 2 || V == 4 || V == 5) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
}
int main() {
  const long M = 802816; const int C = 432, nchunks = (C + 63) / 64;
  const size_t bytes = (size_t)nchunks * 4 * M * 32;
  u32x4* y; hipMalloc(&y, bytes * 2); hipMemset(y, 0, bytes * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int tpi : {16, 69}) {
    const long mtiles = M / 16, nranges = (mtiles + tpi - 1) / tpi, items = nranges * nchunks;
    const int grid = (int)((items + 3) / 4);
#define RUN(V, W)                                                                                      \
    { for (int i = 0; i < 2; ++i) k<V, W><<<grid, 256>>>(y + (i & 1) * (bytes / 16), M, nchunks, tpi); \
      hipDeviceSynchronize(); hipEventRecord(e0);                                                      \
      for (int i = 0; i < 10; ++i) k<V, W><<<grid, 256>>>(y + (i & 1) * (bytes / 16), M, nchunks, tpi); \
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);         \
      printf("tiles/item %3d grid %6d V%d W%3d: %.1f us  %.2f TB/s\n", tpi, grid, V, W, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12); }
    RUN(0, 0) RUN(6, 16) RUN(6, 32) RUN(6, 64) RUN(6, 128) RUN(7, 0) RUN(7, 32) RUN(7, 64) RUN(7, 128)
  }
  return 0;
}

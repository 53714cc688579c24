// Experiment (not product): does the 16-byte-piece store order of k_gemm_nt_cs cost write bandwidth?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/stpat.hip -o tools/probe/stpat && tools/probe/stpat
// Tensor: slab-major bf16 [C/16][M][16] (32-byte rows per slab).  A wave owns 4 slabs (64 channels) and a range of 16-row tiles,
// lane (q, j) as in the kernel.  PAT 0: two stores per tile, lane writes row j, half h (16-byte pieces at a 32-byte stride: every
// instruction half-fills its lines).  PAT 1: store s writes row (j>>1)+8s, half j&1 (256 contiguous bytes per quarter wave).
// PAT 2: like 0 but reading (the z operand of the masked input-gradient GEMM).  PAT 3: like 1, reading.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int PAT>
__global__ __launch_bounds__(256) void k(u32x4* __restrict__ y, long M, int nchunks, int tiles_per_item) {
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long mtiles = M / 16, nranges = (mtiles + tiles_per_item - 1) / tiles_per_item;
  if (item >= nranges * nchunks) return;
  const int chunk = item % nchunks; const long range = item / nchunks;
  const long ss16 = M * 2;   // slab stride in 16-byte units
  u32x4* base = y + (long)(chunk * 4 + q) * ss16;
  long mt = range * tiles_per_item, me = mt + tiles_per_item < mtiles ? mt + tiles_per_item : mtiles;
  u32x4 acc = {0, 0, 0, 0};
  for (; mt < me; ++mt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      long row; int half;
      if (PAT & 1) { row = mt * 16 + (j >> 1) + 8 * s; half = j & 1; } else { row = mt * 16 + j; half = s; }
      if (PAT < 2) base[row * 2 + half] = u32x4{(unsigned)mt, (unsigned)s, 3u, 4u};
      else acc += base[row * 2 + half];
    }
  }
  if (PAT >= 2 && acc[0] == 0x12345678u) y[0] = acc;
}
int main() {
  const long M = 802816; const int C = 432, nchunks = (C + 63) / 64;   // 7 chunks -> 448 channels allocated
  const size_t bytes = (size_t)nchunks * 4 * M * 32;
  u32x4* y; hipMalloc(&y, bytes); hipMemset(y, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int tpi : {8, 16, 32, 64}) {
    const long mtiles = M / 16, nranges = (mtiles + tpi - 1) / tpi, items = nranges * nchunks;
    const int grid = (int)((items + 3) / 4);
#define RUN(P)                                                                                         \
    { for (int i = 0; i < 2; ++i) k<P><<<grid, 256>>>(y, M, nchunks, tpi);                             \
      hipDeviceSynchronize(); hipEventRecord(e0);                                                      \
      for (int i = 0; i < 10; ++i) k<P><<<grid, 256>>>(y, M, nchunks, tpi);                            \
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);         \
      printf("tiles/item %3d grid %6d pat %d: %.1f us  %.2f TB/s\n", tpi, grid, P, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12); }
    RUN(0) RUN(1) RUN(2) RUN(3)
  }
  return 0;
}

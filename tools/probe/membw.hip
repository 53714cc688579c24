// Experiment (not product): streaming rates of MI355X for the access orders the depthwise / pointwise kernels can choose.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/membw.hip -o tools/probe/membw.bin && tools/probe/membw.bin
// 1. linear float4 streams (read, write, copy, 3 reads + 1 write), grid-stride vs one contiguous chunk per workgroup
// 2. NHWC [M][ld] bf16 with channel slabs of CB channels (CB*2 bytes of every pixel row): read / copy / 3r1w per slab width,
//    workgroups walking contiguous row ranges (persistent) -- what a channel-slab depthwise kernel can at best reach
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// mode: 0 read, 1 write, 2 copy, 3 = 3 reads + 1 write
template <int MODE, int CHUNKED>
__global__ __launch_bounds__(256) void k_lin(const u32x4* __restrict__ a, const u32x4* __restrict__ b, const u32x4* __restrict__ c,
                                             u32x4* __restrict__ y, long n) {
  u32x4 acc = {0, 0, 0, 0};
  long beg, end, step;
  if (CHUNKED) { const long per = (n + gridDim.x - 1) / gridDim.x; beg = blockIdx.x * per + threadIdx.x; end = min(n, (blockIdx.x + 1) * per); step = 256; }
  else { beg = (long)blockIdx.x * 256 + threadIdx.x; end = n; step = (long)gridDim.x * 256; }
  for (long i = beg; i < end; i += step * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long j = i + u * step;
      v[u] = u32x4{1, 2, 3, 4};
      if (j < end && MODE != 1) { v[u] = a[j]; if (MODE == 3) { v[u] += b[j]; v[u] += c[j]; } }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long j = i + u * step;
      if (j < end) { if (MODE == 0) acc += v[u]; else y[j] = v[u]; }
    }
  }
  if (MODE == 0 && acc[0] == 0x12345678u) y[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_slab(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                              const unsigned short* __restrict__ c, unsigned short* __restrict__ y, long M, int ld, int CB,
                                              int nslabs, int nworkers) {
  const int CG = CB / 8;
  const int b_xcd = blockIdx.x & 7, b_local = blockIdx.x >> 3;
  const int slab = b_local % nslabs, worker = (b_local / nslabs) * 8 + b_xcd;
  if (worker >= nworkers) return;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG, RPB = 256 / CG;
  const long r_beg = M * worker / nworkers, r_end = M * (worker + 1) / nworkers;
  const long coff = (long)slab * CB + cg * 8;
  u32x4 acc = {0, 0, 0, 0};
  for (long r = r_beg + rl; r < r_end; r += (long)RPB * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = r + (long)u * RPB;
      v[u] = u32x4{0, 0, 0, 0};
      if (rr < r_end) {
        v[u] = *reinterpret_cast<const u32x4*>(a + rr * ld + coff);
        if (MODE == 3) { v[u] += *reinterpret_cast<const u32x4*>(b + rr * ld + coff); v[u] += *reinterpret_cast<const u32x4*>(c + rr * ld + coff); }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = r + (long)u * RPB;
      if (rr < r_end) { if (MODE == 0) acc += v[u]; else *reinterpret_cast<u32x4*>(y + rr * ld + coff) = v[u]; }
    }
  }
  if (MODE == 0 && acc[0] == 0x12345678u) y[0] = 1;
}

int main() {
  const size_t bytes = 768ull << 20;   // per stream, > Infinity Cache
  void *a, *b, *c, *y;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&y, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 3, bytes); hipMemset(y, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long n16 = bytes / 16;
  const char* mname[] = {"read", "write", "copy", "3r1w"};
  const double mbytes[] = {1, 1, 2, 4};
#define TIME(label, launch, moved)                                                                \
  {                                                                                               \
    for (int i = 0; i < 2; ++i) { launch; }                                                       \
    hipDeviceSynchronize(); hipEventRecord(e0);                                                   \
    for (int i = 0; i < 5; ++i) { launch; }                                                       \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                  \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;                                          \
    printf("%-46s %8.3f ms  %6.0f GB/s (all bytes moved)\n", label, ms, (moved) / ms / 1e6);      \
  }
  char lab[128];
  for (int grid : {2048, 8192}) {
#define LIN(MODE, CH)                                                                                                         \
    snprintf(lab, sizeof lab, "linear %s %s grid %d", mname[MODE], CH ? "chunk/WG" : "grid-stride", grid);                     \
    TIME(lab, (k_lin<MODE, CH><<<dim3(grid), dim3(256)>>>((const u32x4*)a, (const u32x4*)b, (const u32x4*)c, (u32x4*)y, n16)), bytes * mbytes[MODE])
    LIN(0, 0) LIN(0, 1) LIN(1, 0) LIN(1, 1) LIN(2, 0) LIN(2, 1) LIN(3, 0) LIN(3, 1)
  }
  // channel slabs: hidden tensor of f3-f5 (56x56, 432 channels) at bs 256 = 694 MB
  const long M = 256L * 56 * 56;
  for (int ld : {144, 432}) {
    const long need = M * ld * 2;
    if ((size_t)need > bytes) { printf("skip ld %d\n", ld); continue; }
    for (int CB : {16, 32, 64}) {
      if (ld % CB && CB != 16) { if (ld != 144 || CB != 64) { /* ragged last slab: ignore tail */ } }
      const int nslabs = ld / CB;
      for (int percu : {4, 8}) {
        const int nworkers = 256 * percu / nslabs;
        const int grid = (nworkers + 7) / 8 * 8 * nslabs;
        const double moved1 = (double)M * CB * nslabs * 2;
#define SLAB(MODE)                                                                                                              \
        snprintf(lab, sizeof lab, "slab %s ld %d CB %d (%d slabs) %d/CU", mname[MODE], ld, CB, nslabs, percu);                     \
        TIME(lab, (k_slab<MODE><<<dim3(grid), dim3(256)>>>((const unsigned short*)a, (const unsigned short*)b, \
                                      (const unsigned short*)c, (unsigned short*)y, M, ld, CB, nslabs, nworkers)), moved1 * mbytes[MODE])
        SLAB(0) SLAB(2) SLAB(3)
      }
    }
  }
  return 0;
}

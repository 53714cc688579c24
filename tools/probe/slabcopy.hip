// Experiment (not product): what streaming rate does a channel-slab access pattern reach on MI355X?
// y[m][c] = x[m][c] for an NHWC-like [M][ld] bf16 matrix; a workgroup owns CB channels (CB*2 bytes of every row) and a
// contiguous range of rows; lanes load 16 bytes.  Same XCD-aware slab placement as the depthwise kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/slabcopy.hip -o tools/probe/slabcopy && tools/probe/slabcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int UNROLL>
__global__ __launch_bounds__(256) void k_slabcopy(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, long M, int ld,
                                                  int CB, int nslabs, int nworkers, int xcd_aware, int do_write) {
  const int CG = CB / 8;
  int slab, worker;
  if (xcd_aware) {
    const int b_xcd = blockIdx.x & 7, b_local = blockIdx.x >> 3;
    slab = b_local % nslabs; worker = (b_local / nslabs) * 8 + b_xcd;
  } else {
    slab = blockIdx.x % nslabs; worker = blockIdx.x / nslabs;
  }
  if (worker >= nworkers) return;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG, RPB = 256 / CG;  // rows per block-iteration
  const long r_beg = M * worker / nworkers, r_end = M * (worker + 1) / nworkers;
  const long coff = (long)slab * CB + cg * 8;
  u32x4 acc = {0, 0, 0, 0};
  for (long r = r_beg + rl; r < r_end; r += (long)RPB * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long rr = r + (long)u * RPB;
      v[u] = u32x4{0, 0, 0, 0};
      if (rr < r_end && do_write != 3) v[u] = *reinterpret_cast<const u32x4*>(x + rr * ld + coff);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long rr = r + (long)u * RPB;
      if (do_write == 1) { if (rr < r_end) *reinterpret_cast<u32x4*>(y + rr * ld + coff) = v[u]; }
      else if (do_write == 2) { if (rr < r_end) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(y + rr * ld + coff)); }
      else if (do_write == 3) { if (rr < r_end) { u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + rr * ld + coff)); __builtin_nontemporal_store(t + v[u], reinterpret_cast<u32x4*>(y + rr * ld + coff)); } }
      else acc += v[u];
    }
  }
  if (!do_write && acc[0] == 0x12345678u) y[0] = 1;
}

int main() {
  const long M = 256L * 56 * 56;
  const int C = 144, ld = 144;
  unsigned short *x, *y;
  // several buffers in rotation so that the 256 MiB Infinity Cache does not serve the reads
  const int NBUF = 6;
  const size_t bytes = (size_t)M * ld * 2;
  hipMalloc(&x, bytes * NBUF); hipMalloc(&y, bytes * NBUF);
  hipMemset(x, 1, bytes * NBUF); hipMemset(y, 0, bytes * NBUF);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("M=%ld C=%d  (%.0f MB per tensor)\n", M, C, bytes / 1e6);
  for (int do_write : {1, 2, 3, 0})
  for (int xcd = 1; xcd >= 1; --xcd)
  for (int CB : {16, 144})
  for (int percu : {4, 8})
  for (int unroll : {4}) {
    if (CB == 48 && 256 % (CB / 8)) continue;
    if (256 % (CB / 8)) { /* CG must divide 256: 144/8=18 does not -> use 128-thread rows */ }
    int cb = CB; if (cb == 144) cb = 128;  // 128 channels = two full lines (tail 16 channels ignored)
    if (cb == 48) cb = 64;
    const int nslabs = (cb == 128) ? 1 : (cb == 64 ? 2 : C / cb);
    int nworkers = 256 * percu / nslabs;
    const int grid = (nworkers + 7) / 8 * 8 * nslabs;
    auto launch = [&](int it) {
      const unsigned short* xi = x + (size_t)(it % NBUF) * M * ld; unsigned short* yi = y + (size_t)(it % NBUF) * M * ld;
      if (unroll == 1) hipLaunchKernelGGL(k_slabcopy<1>, dim3(grid), dim3(256), 0, 0, xi, yi, M, ld, cb, nslabs, nworkers, xcd, do_write);
      else hipLaunchKernelGGL(k_slabcopy<4>, dim3(grid), dim3(256), 0, 0, xi, yi, M, ld, cb, nslabs, nworkers, xcd, do_write);
    };
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize();
    const int IT = 12;
    hipEventRecord(e0);
    for (int i = 0; i < IT; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= IT;
    const double moved = (double)M * cb * nslabs * 2 * (do_write ? 2 : 1);
    const char* names[] = {"read", "copy", "copy-nt-store", "copy-nt-both"};
    printf("%s xcd=%d CB=%3d nslabs=%d percu=%d unroll=%d : %.3f ms  %.0f GB/s\n", names[do_write], xcd, cb, nslabs, percu, unroll,
           ms, moved / ms / 1e6);
  }
  return 0;
}

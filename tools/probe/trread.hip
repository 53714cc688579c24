// Probe (gfx950): semantics of ds_read_b64_tr_b16, the LDS transpose read, as the weight-gradient GEMM needs it (rows = GEMM k).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/trread.hip -o tools/probe/trread.bin && tools/probe/trread.bin
// LDS holds a [32][16] bf16 tile, row-major, value(row, col) = 100 * row + col.  Every lane passes the address of 4 contiguous
// elements: lane l of 16-lane group g = l >> 4, sub-lane t = l & 15, points at row (8 g + t / 4 + ROW_OFF), columns 4 (t % 4) .. + 3.
// Printed: what every lane received.  Expectation (MFMA operand, k along rows): lane (c = l & 15, g) element e = tile[8 g + e + ROW_OFF][c].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[32 * 16];
  for (int i = threadIdx.x; i < 512; i += 64) tile[i] = (bf16_t)(float)(100 * (i / 16) + (i % 16));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, t = l & 15;
  for (int half = 0; half < 2; ++half) {
    const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(tile + (8 * g + 4 * half + t / 4) * 16 + 4 * (t % 4)));
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int e = 0; e < 4; ++e) out[(half * 64 + l) * 4 + e] = (float)v[e];
  }
}
int main() {
  float* d; hipMalloc(&d, 2 * 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int half = 0; half < 2; ++half)
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) {
        const float want = 100.f * (8 * (l >> 4) + 4 * half + e) + (l & 15);
        if (h[(half * 64 + l) * 4 + e] != want) ++bad;
      }
  printf("ds_read_b64_tr_b16: %d of 512 values differ from tile[8 g + 4 half + e][lane & 15]\n", bad);
  for (int l = 0; l < 20; ++l) printf("lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}

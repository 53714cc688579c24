// Probe: do LDS reads at addresses that are not a multiple of their size work on gfx950 (SH_MEM alignment mode), and what do they cost?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/ldsmis.hip -o tools/probe/ldsmis && tools/probe/ldsmis
// Each lane reads 16 / 8 / 4 bytes at byte offset (32 * lane + 2 * shift) of an LDS array of ushorts holding their own index.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int W>
__global__ void k_probe(unsigned* out, int shift, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned short s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (unsigned short)i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)s);
  const unsigned ad = base + 32u * (threadIdx.x & 63) + 2u * shift + 2048u * (threadIdx.x >> 6);
  u32x4 acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v = {0, 0, 0, 0};
    if (W == 16) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
    if (W == 8) { u32x2 t; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(ad) : "memory"); v[0] = t[0]; v[1] = t[1]; }
    if (W == 4) { unsigned t; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(ad) : "memory"); v[0] = t; }
    acc += v;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = iters == 1 ? acc[i] : acc[i] / iters;
}

template <int W> void run(int shift) {
  unsigned* d; unsigned long long* c;
  hipMalloc(&d, 256 * 16); hipMalloc(&c, 8);
  hipLaunchKernelGGL(k_probe<W>, dim3(1), dim3(256), 0, 0, d, shift, 1, c);
  std::vector<unsigned> h(1024);
  hipError_t e = hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256 && e == hipSuccess; ++t)
    for (int i = 0; i < W / 4; ++i) {
      const unsigned first = 16 * (t & 63) + shift + 1024 * (t >> 6) + 2 * i;
      const unsigned expect = (first & 0xffff) | ((first + 1) << 16);
      if (h[t * 4 + i] != expect) { if (bad < 2) printf("    lane %d dword %d: got %08x expect %08x\n", t, i, h[t * 4 + i], expect); ++bad; }
    }
  hipLaunchKernelGGL(k_probe<W>, dim3(1), dim3(256), 0, 0, d, shift, 2000, c);
  unsigned long long cy = 0; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
  printf("width %2d  shift %d halves (byte offset %% %d = %d): %s (%d wrong dwords, err %d)   %.1f cycles per dependent read (4 waves)\n", W, shift, W,
         (2 * shift) % W, bad == 0 && e == hipSuccess ? "OK" : "WRONG", bad, (int)e, cy / 2000.0);
  hipFree(d); hipFree(c);
}

int main() {
  for (int shift = 0; shift < 8; ++shift) run<16>(shift);
  for (int shift = 0; shift < 4; ++shift) run<8>(shift);
  for (int shift = 0; shift < 2; ++shift) run<4>(shift);
  return 0;
}

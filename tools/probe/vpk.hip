// Micro-benchmark (gfx950): issue cost of the instructions the depthwise tap rows are made of.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/vpk.hip -o gpurun_out/vpk && gpurun_out/vpk
// Each variant: a loop of 32 iterations x (block of 56 instructions), one wave per SIMD or two, cycles per instruction from
// s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float seed, int iters, const float* lds_init) {
  __shared__ f32x2 s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = f32x2{seed + i, seed};
  __syncthreads();
  f32x2 a[14], d[13], w = f32x2{seed, seed * 0.5f};
  float ws0 = __builtin_amdgcn_readfirstlane(__float_as_int(seed)) * 1.0f, ws1 = ws0 * 0.25f;
  f32x2 wsv = f32x2{ws0, ws1};
#pragma unroll
  for (int i = 0; i < 14; ++i) a[i] = f32x2{seed * i, seed + i};
#pragma unroll
  for (int i = 0; i < 13; ++i) d[i] = f32x2{seed - i, seed * (i + 1)};
  const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) char*)&s[(threadIdx.x & 63) * 7 + (threadIdx.x >> 6) * 448]);
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // 56 v_pk_fma_f32, VGPR operands, 14 independent accumulators
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 14; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(d[i % 13]), "v"(w));
    } else if (MODE == 1) {   // same with an SGPR-pair operand
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 14; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(d[i % 13]), "s"(wsv));
    } else if (MODE == 2) {   // 112 v_fma_f32
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 14; ++i) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][0]) : "v"(d[i % 13][0]), "v"(w[0]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][1]) : "v"(d[i % 13][1]), "v"(w[1]));
        }
    } else if (MODE == 3) {   // the tap row: 13 ds_read_b64 + wait + 98 pk_fma (7 with SGPR w) + 14 readlane
      f32x2 dy[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dy[i]) : "v"(addr), "i"(i * 8));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[t]) : "v"(dy[t + 6 - kx]), "s"(wsv));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[7 + kx]) : "v"(d[t]), "v"(dy[t + 6 - kx]));
        }
    } else if (MODE == 4) {   // 13 ds_read_b64 only (conflict-free pattern), waited
      f32x2 dy[13];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 13; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dy[i]) : "v"(addr), "i"(i * 8 + r * 128));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 13; ++i) asm volatile("" ::"v"(dy[i]));
      }
    } else if (MODE == 5) {   // 56 v_readlane_b32
#pragma unroll
      for (int i = 0; i < 56; ++i) {
        int sv;
        asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(sv) : "v"(a[i % 14][0]), "i"(i));
        asm volatile("" ::"s"(sv));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 14; ++i) acc += a[i][0] + a[i][1];
  if (acc == 12345.678f) out[1000] = 1;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE> void run(const char* name, int per_iter, int threads) {
  unsigned long long* d;
  hipMalloc(&d, 8192);
  hipMemset(d, 0, 8192);
  const int iters = 64;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.5f, iters, nullptr);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.5f, iters, nullptr);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("%-44s %d wave(s)/SIMD: %.2f cycles per instruction (wave 0), %.2f (last wave)\n", name, threads / 256, (double)h[0] / iters / per_iter,
         (double)h[threads / 64 - 1] / iters / per_iter);
  hipFree(d);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("v_pk_fma_f32 vgpr operands", 56, threads);
    run<1>("v_pk_fma_f32 sgpr-pair operand", 56, threads);
    run<2>("v_fma_f32", 112, threads);
    run<3>("tap row (13 ds_read_b64 + 98 pk_fma), per pk_fma", 98, threads);
    run<4>("ds_read_b64 x13 waited, per read", 52, threads);
    run<5>("v_readlane_b32", 56, threads);
  }
  return 0;
}

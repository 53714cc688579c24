// Fixed-order reductions shared by the kernels that used to accumulate with atomics: per-workgroup partial weight
// gradients (depthwise taps, 1x1 weight-gradient GEMM) are written to a caller-owned workspace and summed here in an
// order that depends only on the launch geometry, so that two runs of the same training step are bit-identical
// (reference semantics: autograd's AccumulateGrad adds ONE complete gradient tensor, utils/rmsprop.py:70-132 then reads it).
#include "common.h"

namespace atomnas {

// block = 32 elements x 8 part-groups; part-group pg sums parts pg, pg+8, ... (4 independent accumulators, combined in a
// fixed tree), then the 8 group sums are added in order 0..7.
__global__ __launch_bounds__(256) void k_reduce_parts(const float* __restrict__ part, long part_stride, int parts, long n,
                                                      float* __restrict__ out, int inner, long s_outer, long s_inner) {
  __shared__ float s_acc[8][32];
  const int el = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const long e = (long)blockIdx.x * 32 + el;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < n) {
    int r = pg;
    for (; r + 24 < parts; r += 32) {
      a0 += part[(long)r * part_stride + e];
      a1 += part[(long)(r + 8) * part_stride + e];
      a2 += part[(long)(r + 16) * part_stride + e];
      a3 += part[(long)(r + 24) * part_stride + e];
    }
    for (; r < parts; r += 8) a0 += part[(long)r * part_stride + e];
  }
  s_acc[pg][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pg == 0 && e < n) {
    float t = s_acc[0][el];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += s_acc[g][el];
    const long o = (e / inner) * s_outer + (e % inner) * s_inner;
    out[o] += t;
  }
}

int reduce_parts(const float* part, long part_stride, int parts, long n, float* out, int inner, long s_outer, long s_inner,
                 hipStream_t st) {
  if (n <= 0 || parts <= 0) return 0;
  const long blocks = (n + 31) / 32;
  hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)blocks), dim3(256), 0, st, part, part_stride, parts, n, out, inner, s_outer,
                     s_inner);
  return check_launch("reduce_parts");
}

}  // namespace atomnas

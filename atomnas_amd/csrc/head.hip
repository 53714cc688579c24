// Network head and tail around the atomic blocks: stem im2col (the 3x3/s2 stem of models/mobilenet_supernet.py:126-132
// is run as a K=27 GEMM through pwconv.hip) and the label-smoothed cross entropy with top-k bookkeeping
// (utils/optim.py:180-207, common.py:67-80) done on device without host synchronisation.
#include "common.h"

namespace atomnas {

// col[m][ci*9 + ky*3 + kx] = img[n][ci][2*ho+ky-1][2*wo+kx-1] (zero outside), m = (n*Ho + ho)*Wo + wo, columns 27..ld-1 zero.
// img is NCHW fp32 (what the reference's loader hands to the model), col has storage type T.
template <typename T>
__global__ __launch_bounds__(256) void k_im2col_stem(const float* __restrict__ img, T* __restrict__ col, int ld, int N, int H, int W,
                                                     int Ho, int Wo) {
  const long total = (long)N * Ho * Wo;
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < total; m += (long)gridDim.x * 256) {
    const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho), n = (int)(m / ((long)Wo * Ho));
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float* plane = img + ((long)n * 3 + ci) * H * W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int hi = 2 * ho + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int wi = 2 * wo + kx - 1;
          if (hi >= 0 && hi < H && wi >= 0 && wi < W) v[ci * 9 + ky * 3 + kx] = plane[(long)hi * W + wi];
        }
      }
    }
    T* o = col + m * ld;
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = v[8 * g8 + e];
      VecIO<T, 8>::store(o + 8 * g8, t);
    }
  }
}

// One wave per sample.  loss_i = -sum_j ((1-eps)[j==y] + eps/K) * log_softmax(x)_j ; dlogits = (softmax - target)/B * gscale.
template <typename T>
__global__ __launch_bounds__(256) void k_ce_smooth(const float* __restrict__ logits, int ldl, const long long* __restrict__ target,
                                                   float eps, int B, int K, float* __restrict__ loss_per_sample,
                                                   T* __restrict__ dlogits, int ldd, float gscale,
                                                   int* __restrict__ topk_correct /*[2]: top1, top5*/) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* x = logits + (long)b * ldl;
  const int y = (int)target[b];
  float mx = -INFINITY;
  for (int j = lane; j < K; j += 64) mx = fmaxf(mx, x[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float se = 0.f, sx = 0.f;
  const float xy = x[y];
  int rank = 0;
  for (int j = lane; j < K; j += 64) {
    const float v = x[j];
    se += expf(v - mx);
    sx += v;
    rank += (v > xy) ? 1 : 0;
  }
  se = wave_sum(se);
  sx = wave_sum(sx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rank += __shfl_xor(rank, o, 64);
  const float lse = mx + logf(se);
  // sum_j logp_j = sx - K*lse ; logp_y = xy - lse
  const float loss = -(1.f - eps) * (xy - lse) - (eps / (float)K) * (sx - (float)K * lse);
  if (lane == 0) {
    if (loss_per_sample) loss_per_sample[b] = loss;
    if (topk_correct) {   // integer atomics: exact and order-independent
      if (rank < 1) atomicAdd(&topk_correct[0], 1);
      if (rank < 5) atomicAdd(&topk_correct[1], 1);
    }
  }
  if (dlogits) {
    T* d = dlogits + (long)b * ldd;
    const float invB = gscale / (float)B;
    for (int j = lane; j < ldd; j += 64) {
      float g = 0.f;
      if (j < K) {
        const float p = expf(x[j] - lse);
        const float t = eps / (float)K + ((j == y) ? (1.f - eps) : 0.f);
        g = (p - t) * invB;
      }
      d[j] = from_f32<T>(g);
    }
  }
}

// column sums of a [M, C] tensor of storage type T into fp32 (classifier bias gradient): one thread per column, rows in
// order (M is the batch size here), so the result is bit-reproducible
template <typename T>
__global__ __launch_bounds__(256) void k_colsum(const T* __restrict__ x, int ld, float* __restrict__ out, long M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long m = 0;
  for (; m + 3 < M; m += 4) {
    s0 += to_f32(x[m * ld + c]);
    s1 += to_f32(x[(m + 1) * ld + c]);
    s2 += to_f32(x[(m + 2) * ld + c]);
    s3 += to_f32(x[(m + 3) * ld + c]);
  }
  for (; m < M; ++m) s0 += to_f32(x[m * ld + c]);
  out[c] += (s0 + s1) + (s2 + s3);
}

}  // namespace atomnas

using namespace atomnas;

extern "C" int atomnas_im2col_stem(const float* img, void* col, int ld, int N, int H, int W, int dtype, void* stream) {
  ATOMNAS_REQUIRE(img && col && ld >= 32 && ld % 8 == 0 && N > 0 && H > 0 && W > 0, "im2col_stem: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_im2col_stem<float>, dim3((unsigned)blocks), dim3(256), 0, st, img, (float*)col, ld, N, H, W, Ho, Wo);
  else
    hipLaunchKernelGGL(k_im2col_stem<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, img, (bf16_t*)col, ld, N, H, W, Ho, Wo);
  return check_launch("im2col_stem");
}

extern "C" int atomnas_ce_smooth(const float* logits, int ldl, const long long* target, float eps, int B, int K, float* loss_per_sample,
                                 void* dlogits, int ldd, float gscale, int* topk_correct, int dtype, void* stream) {
  ATOMNAS_REQUIRE(logits && target && B > 0 && K > 0 && ldl >= K, "ce_smooth: bad arguments");
  ATOMNAS_REQUIRE(!dlogits || ldd >= K, "ce_smooth: bad gradient pitch");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_ce_smooth<float>, dim3((B + 3) / 4), dim3(256), 0, st, logits, ldl, target, eps, B, K, loss_per_sample,
                       (float*)dlogits, ldd, gscale, topk_correct);
  else
    hipLaunchKernelGGL(k_ce_smooth<bf16_t>, dim3((B + 3) / 4), dim3(256), 0, st, logits, ldl, target, eps, B, K, loss_per_sample,
                       (bf16_t*)dlogits, ldd, gscale, topk_correct);
  return check_launch("ce_smooth");
}

extern "C" int atomnas_colsum(const void* x, int ld, float* out, long M, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && out && M > 0 && C > 0 && ld >= C, "colsum: bad arguments");
  dim3 grid((C + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32) hipLaunchKernelGGL(k_colsum<float>, grid, dim3(256), 0, st, (const float*)x, ld, out, M, C);
  else hipLaunchKernelGGL(k_colsum<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ld, out, M, C);
  return check_launch("colsum");
}

// Expand backward WITHOUT the raw expand output E (gfx950, bf16): the inp x inp sized pieces.
//
// The atomic block (models/mobilenet_base.py:316-336, 371-382) expands its narrow input x [M][inp] to the 6x wider E = x We^T and
// normalises it; the backward of that BatchNorm is dE = c1*h + c2*E + c3 (h: the masked gradient of the activated hidden tensor).
// The expand convolution's backward then needs dE in two wide GEMMs (dX = dE We, dWe = dE^T x), which used to read BOTH hidden
// streams h and E.  With E = x We^T the c2 / c3 terms collapse to inp x inp sized corrections:
//       dX  = (c1*h) We + x M + v,          M = We^T diag(c2) We,  v = c3^T We
//       dWe = diag(c1) h^T x + diag(c2) We (X^T X) + c3 (sum x)^T
// so the wide GEMMs read h alone (c1 as their scale prologue: atomnas_expand_bwd with e = NULL, or atomnas_pw_gemm_nt /
// atomnas_pw_gemm_tn with the BNRELU prologue) and this file supplies
//   * atomnas_gram:      G = X^T X and sx = sum x from one pass over the narrow tensor,
//   * atomnas_xb_coeffs: M (packed as a GEMM weight), v, and the last two terms of dWe.
// (The same algebra with the expand recomputed INSIDE the depthwise kernels -- E never in HBM at all -- was built and measured
// in round 4 and lost: csrc/experimental/, DESIGN.md.)  No atomics; all reductions in a fixed order.
#include "common.h"
#include <cstdlib>

namespace atomnas {

// ------------------------------------------------------------------------------------------------- Gram matrix of the block input
// G = X^T X (inp x inp) and sx = sum_m x_m of the narrow block input x [M][ldx] (bf16): per-workgroup partials [G | sx] over a range
// of 128-pixel tiles.  A tile is staged in LDS as fp32 with one more column of ones (so that sx is column inp of the product); a
// thread owns a 4 x 4 block of the product and one of `nsplit` pixel subsets of the tile (two 16-byte LDS reads per 16 FMAs); the
// subsets are added in order at the end, the workgroup partials in workgroup order by k_gram_reduce -- fixed order, no atomics.
// inp <= 64, a multiple of 8.
__global__ __launch_bounds__(256) void k_gram_part(const bf16_t* __restrict__ x, int ldx, long M, int inp, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float s_t[];   // [128][inp + 4], then the reduction buffer [nsplit][nitems][16]
  const int tid = threadIdx.x;
  const int pitch = inp + 4, nr = inp >> 2, nc = nr + 1, nitems = nr * nc;
  const int nsplit = 256 / nitems > 0 ? (256 / nitems > 16 ? 16 : 256 / nitems) : 1;   // inp <= 56: 256 / nitems >= 1; inp = 64: 272 items, two passes
  const int npass = (nitems * nsplit + 255) / 256;
  f32x4 acc[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[u][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long ntiles = (M + 127) / 128;
  const long t_beg = blockIdx.x * ntiles / gridDim.x, t_end = (blockIdx.x + 1) * ntiles / gridDim.x;
  const int npc = 128 * (inp >> 3);   // 16-byte pieces per tile
  // the pieces of a tile (at most 4 per thread: inp <= 64) are fetched one tile ahead
  bf16x8 pf[4];
  auto fetch = [&](long t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pc = tid + 256 * u;
      const int r = pc / (inp >> 3), cg = pc - r * (inp >> 3);
      const long row = t * 128 + r;
      const bool ok = pc < npc && row < M;
      pf[u] = *reinterpret_cast<const bf16x8*>(x + (ok ? row * ldx + cg * 8 : 0));
    }
  };
  if (t_beg < t_end) fetch(t_beg);
  for (long t = t_beg; t < t_end; ++t) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pc = tid + 256 * u;
      if (pc < npc) {
        const int r = pc / (inp >> 3), cg = pc - r * (inp >> 3);
        const bool ok = t * 128 + r < M;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ok ? (float)pf[u][e] : 0.f;
        VecIO<float, 8>::store(s_t + r * pitch + cg * 8, v);
        if (cg == 0) *reinterpret_cast<f32x4*>(s_t + r * pitch + inp) = f32x4{ok ? 1.f : 0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
    fetch(t + 1 < t_end ? t + 1 : t);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int w = tid + 256 * u;
      if (u < npass && w < nitems * nsplit) {
        const int it = w % nitems, sp = w / nitems;
        const int bi = it / nc, bj = it - bi * nc;
        const float* pa = s_t + 4 * bi;
        const float* pb = s_t + 4 * bj;
        for (int pp = sp; pp < 128; pp += nsplit) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(pa + pp * pitch);
          const f32x4 b = *reinterpret_cast<const f32x4*>(pb + pp * pitch);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][r] += f32x4{a[r], a[r], a[r], a[r]} * b;
        }
      }
    }
  }
  // the pixel subsets of an item are added in subset order
  __syncthreads();
  float* red = s_t;   // [nsplit][nitems][16] <= 256 * 2 * 16 floats <= the tile buffer (128 * (inp + 4), inp >= 8: host side checks)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int w = tid + 256 * u;
    if (u < npass && w < nitems * nsplit) {
      const int it = w % nitems, sp = w / nitems;
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(red + ((long)sp * nitems + it) * 16 + 4 * r) = acc[u][r];
    }
  }
  __syncthreads();
  float* o = ws + (long)blockIdx.x * (inp * inp + inp);
  for (int e = tid; e < nitems * 16; e += 256) {
    const int it = e >> 4, r = (e >> 2) & 3, c = e & 3;
    float a = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) a += red[((long)sp * nitems + it) * 16 + 4 * r + c];
    const int bi = it / nc, bj = it - bi * nc;
    const int i = 4 * bi + r, jj = 4 * bj + c;
    if (jj < inp) o[i * inp + jj] = a;
    else if (jj == inp) o[inp * inp + i] = a;
  }
}
// one wave per output element: lane l adds the partials l, l + 64, ... in order, then a fixed-order butterfly over the lanes
__global__ __launch_bounds__(256) void k_gram_reduce(const float* __restrict__ ws, int parts, int inp, float* __restrict__ gram,
                                                     float* __restrict__ sx) {
  const int n = inp * inp + inp;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= n) return;
  float a = 0.f;
  for (int r = lane; r < parts; r += 64) a += ws[(long)r * n + e];
  a = wave_sum(a);
  if (lane == 0) {
    if (e < inp * inp) gram[e] = a; else sx[e - inp * inp] = a;
  }
}

// The inp x inp sized corrections of the expand backward without E (see the file header):
//   mp[n][k]      = bf16( sum_c c2_c W[c][n] W[c][k] )          packed for atomnas_pw_gemm_nt ([inp rounded up to 64][ldm], zero padded by the caller)
//   vb[n]         = sum_c c3_c W[c][n]                          (its bias vector)
//   dwe[c*inp+k] += c2_c sum_j W[c][j] G[j][k] + c3_c sx[k]
// Workgroups 0 .. inp-1 own one row n of M (and v[n]): 256 threads = KP columns x S channel subsets, the subsets added in order;
// the remaining workgroups own 256 elements of dwe each.  Fixed-order sums.
__global__ __launch_bounds__(256) void k_xb_coeffs(const float* __restrict__ c2, const float* __restrict__ c3, const bf16_t* __restrict__ wexp,
                                                   int ldwe, const float* __restrict__ gram, int ldg, const float* __restrict__ sx,
                                                   int inp, int C, bf16_t* __restrict__ mp, int ldm,
                                                   float* __restrict__ vb, float* __restrict__ dwe) {
  __shared__ float s_p[256 + 64];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < inp) {
    const int n = blockIdx.x;
    int KP = 8;
    while (KP < inp) KP *= 2;   // 8 .. 64
    const int S = 256 / KP;
    const int k = tid % KP, sub = tid / KP;
    float a = 0.f, va = 0.f;
    for (int c = sub; c < C; c += S) {
      const float wn = (float)wexp[(long)c * ldwe + n];
      if (k < inp) a += c2[c] * wn * (float)wexp[(long)c * ldwe + k];
      if (k == 0) va += c3[c] * wn;
    }
    s_p[sub * KP + k] = a;
    if (k == 0) s_p[256 + sub] = va;
    __syncthreads();
    if (sub == 0 && k < inp) {
      float t = s_p[k];
      for (int q = 1; q < S; ++q) t += s_p[q * KP + k];
      mp[(long)n * ldm + k] = (bf16_t)t;
    }
    if (tid == 0) {
      float t = s_p[256];
      for (int q = 1; q < S; ++q) t += s_p[256 + q];
      vb[n] = t;
    }
    return;
  }
  const long e = (long)(blockIdx.x - inp) * 256 + tid;
  if (e >= (long)C * inp) return;
  const int c = (int)(e / inp), k = (int)(e % inp);
  float a = 0.f;
  for (int jj = 0; jj < inp; ++jj) a += (float)wexp[(long)c * ldwe + jj] * gram[jj * ldg + k];
  dwe[e] += c2[c] * a + c3[c] * sx[k];
}

}  // namespace atomnas

using namespace atomnas;

// Gram matrix G = X^T X [inp][inp] and column sums sx [inp] of the block input x [M][ldx] (bf16; inp <= 64, a multiple of 8).
// ws: caller-owned scratch of ws_floats floats for the per-workgroup partials (inp*inp + inp floats each; at least one).
extern "C" int atomnas_gram(const void* x, int ldx, long M, int inp, float* ws, long ws_floats, float* gram, float* sx, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && ws && gram && sx && M > 0 && inp >= 8 && inp <= 64 && inp % 8 == 0 && ldx >= inp && ldx % 8 == 0 && dtype == DT_BF16,
                  "gram: bad arguments (bf16, inp <= 64 and a multiple of 8)");
  const long ps = (long)inp * inp + inp;
  long parts = 2L * num_cus();
  const long ntiles = (M + 127) / 128;
  if (parts > ntiles) parts = ntiles;
  if (parts > ws_floats / ps) parts = ws_floats / ps;
  ATOMNAS_REQUIRE(parts >= 1, "gram: workspace too small for one partial (%ld floats)", ps);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = (size_t)128 * (inp + 4) * sizeof(float);
  if (lds < (size_t)512 * 16 * sizeof(float)) lds = (size_t)512 * 16 * sizeof(float);   // the end-of-kernel reduction buffer
  hipLaunchKernelGGL(k_gram_part, dim3((unsigned)parts), dim3(256), lds, st, (const bf16_t*)x, ldx, M, inp, ws);
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((ps + 3) / 4)), dim3(256), 0, st, ws, (int)parts, inp, gram, sx);
  return check_launch("gram");
}

// Corrections of the expand backward without E (see k_xb_coeffs); c2 / c3: BatchNorm-backward coefficients of the C hidden channels.
extern "C" int atomnas_xb_coeffs(const float* c2, const float* c3, const void* wexp, int ldwe, const float* gram, int ldg, const float* sx,
                                 int inp, int C, void* mp, int ldm, float* vb, float* dwe, void* stream) {
  ATOMNAS_REQUIRE(c2 && c3 && wexp && gram && sx && mp && vb && dwe && inp > 0 && inp <= 64 && C > 0 && ldg >= inp && ldwe >= inp && ldm >= inp,
                  "xb_coeffs: bad arguments");
  const long blocks = inp + ((long)C * inp + 255) / 256;
  hipLaunchKernelGGL(k_xb_coeffs, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, c2, c3, (const bf16_t*)wexp, ldwe, gram, ldg, sx,
                     inp, C, (bf16_t*)mp, ldm, vb, dwe);
  return check_launch("xb_coeffs");
}


// Depthwise k x k convolution (k in {3,5,7}, stride in {1,2}, pad (k-1)/2) for NHWC activations on gfx950.
//
// Replaces the ATen call behind  nn.Conv2d(hid, hid, k, stride, pad, groups=hid, bias=False)  in the reference's
// atomic block (models/mobilenet_base.py:330-336 via ConvBNReLU :120-142), forward and backward, and fuses the
// surrounding BatchNorm / ReLU passes into the load and store sides so that each activation is streamed once:
//
//   forward : y = dwconv( act(x * in_scale + in_shift) )        + per-channel sum(y), sum(y^2) for the next BN
//   backward: dYraw = c1*g + c2*yraw + c3   (BN-backward of the BN that follows the conv, applied on load)
//             dXa   = dwconv^T(dYraw),  dW += corr(act(x*in_scale+in_shift), dYraw)
//             h     = dXa * [x*in_scale+in_shift > 0]           (ReLU-backward of the producer's activation)
//             + per-channel sum(h), sum(h*x) for the BN-backward of the producer's BN
//
// Work decomposition (HBM-bound, VALU-heavy): a thread owns CV consecutive channels and a column strip of TW
// pixels, and walks down the rows of one image chunk; re-reads of the (k-1) halo rows come from the same CU's
// L1 / the XCD's L2.  Lanes of a wave run along channels first, so a wave reads contiguous NHWC bytes.
#include "common.h"

namespace atomnas {

constexpr __host__ __device__ int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
constexpr __host__ __device__ int pmod(int a, int b) { return ((a % b) + b) % b; }

struct DwGeom {
  int N, H, W, C, Ho, Wo;
  int cvb;      // channel-vectors per block (lanes along channels)
  int pb;       // pixel lanes per block
  int spr;      // column strips per row
  int rh;       // rows per chunk
  int nchunks;  // row chunks per image
};

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int K, int S, int CV, int TW>
__global__ __launch_bounds__(256) void k_dwconv_fwd(const T* __restrict__ x, int ldx, const float* __restrict__ in_scale,
                                                    const float* __restrict__ in_shift, int in_relu,
                                                    const float* __restrict__ w, int ldw, T* __restrict__ y, int ldy,
                                                    float* __restrict__ stats, int stat_ld, DwGeom g) {
  constexpr int P = (K - 1) / 2;
  constexpr int IW = (TW - 1) * S + K;
  __shared__ float s_red[256 * 2];  // [cvb*CV][2] block partials of sum / sumsq (cvb*CV <= 256)

  const int tid = threadIdx.x;
  const int cvl = tid % g.cvb;
  const int pl = tid / g.cvb;
  const int c0 = (blockIdx.y * g.cvb + cvl) * CV;
  const long gp = (long)blockIdx.x * g.pb + pl;
  const long nstrips = (long)g.N * g.nchunks * g.spr;
  const bool active = (pl < g.pb) && (c0 < g.C) && (gp < nstrips);

  if (stats) {
    for (int i = tid; i < g.cvb * CV * 2; i += 256) s_red[i] = 0.f;
    __syncthreads();
  }

  float ssum[CV], ssq[CV];
#pragma unroll
  for (int c = 0; c < CV; ++c) ssum[c] = ssq[c] = 0.f;

  if (active) {
    const int ws = (int)(gp % g.spr);
    const int chunk = (int)((gp / g.spr) % g.nchunks);
    const int n = (int)(gp / ((long)g.spr * g.nchunks));
    const int wo0 = ws * TW;
    const int wi0 = wo0 * S - P;

    float wr[K * K][CV];
#pragma unroll
    for (int t = 0; t < K * K; ++t) VecIO<float, CV>::load(w + (long)t * ldw + c0, wr[t]);
    float sc[CV], sh[CV];
#pragma unroll
    for (int c = 0; c < CV; ++c) { sc[c] = 1.f; sh[c] = 0.f; }
    if (in_scale) { VecIO<float, CV>::load(in_scale + c0, sc); VecIO<float, CV>::load(in_shift + c0, sh); }
    bool cvalid[CV];
#pragma unroll
    for (int c = 0; c < CV; ++c) cvalid[c] = (c0 + c) < g.C;

    const int ho_beg = chunk * g.rh;
    const int ho_end = min(g.Ho, ho_beg + g.rh);
    const T* xn = x + (long)n * g.H * g.W * ldx + c0;
    T* yn = y + (long)n * g.Ho * g.Wo * ldy + c0;

    for (int ho = ho_beg; ho < ho_end; ++ho) {
      float acc[TW][CV];
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int c = 0; c < CV; ++c) acc[t][c] = 0.f;

#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int hi = ho * S + ky - P;
        if (hi < 0 || hi >= g.H) continue;
        const T* xr = xn + (long)hi * g.W * ldx;
        float in[IW][CV];
#pragma unroll
        for (int j = 0; j < IW; ++j) {
          const int wi = wi0 + j;
          if (wi >= 0 && wi < g.W) {
            float v[CV];
            VecIO<T, CV>::load(xr + (long)wi * ldx, v);
#pragma unroll
            for (int c = 0; c < CV; ++c) {
              float a = v[c] * sc[c] + sh[c];
              in[j][c] = in_relu ? fmaxf(a, 0.f) : a;
            }
          } else {
#pragma unroll
            for (int c = 0; c < CV; ++c) in[j][c] = 0.f;
          }
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int c = 0; c < CV; ++c) acc[t][c] += in[t * S + kx][c] * wr[ky * K + kx][c];
      }

      T* yr = yn + (long)ho * g.Wo * ldy;
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const int wo = wo0 + t;
        if (wo < g.Wo) {
          float o[CV];
#pragma unroll
          for (int c = 0; c < CV; ++c) {
            // statistics are taken on the stored (rounded) value so that normalisation is self-consistent
            float r = cvalid[c] ? to_f32(from_f32<T>(acc[t][c])) : 0.f;
            o[c] = r;
            ssum[c] += r;
            ssq[c] += r * r;
          }
          VecIO<T, CV>::store(yr + (long)wo * ldy, o);
        }
      }
    }
  }

  if (stats) {
    if (active) {
#pragma unroll
      for (int c = 0; c < CV; ++c) {
        atomicAdd(&s_red[(cvl * CV + c) * 2 + 0], ssum[c]);
        atomicAdd(&s_red[(cvl * CV + c) * 2 + 1], ssq[c]);
      }
    }
    __syncthreads();
    for (int i = tid; i < g.cvb * CV; i += 256) {
      const int c = blockIdx.y * g.cvb * CV + i;
      if (c < g.C) {
        atomicAdd(&stats[c], s_red[i * 2 + 0]);
        atomicAdd(&stats[stat_ld + c], s_red[i * 2 + 1]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// One thread owns CV channels and a column strip of TW *input* pixels; for every input row it gathers the
// contributing output rows.  For stride 2 only taps with matching parity contribute (static per (t,kx), uniform
// per row for ky).
template <typename T, int K, int S, int CV, int TW>
__global__ __launch_bounds__(256) void k_dwconv_bwd(const T* __restrict__ gup, int ldg, const T* __restrict__ yraw, int ldyr,
                                                    const float* __restrict__ c1, const float* __restrict__ c2,
                                                    const float* __restrict__ c3, const T* __restrict__ x, int ldx,
                                                    const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                    int in_relu, const float* __restrict__ w, int ldw,
                                                    T* __restrict__ h, int ldh, float* __restrict__ dw /*[C][K*K]*/,
                                                    float* __restrict__ stats /*[2][stat_ld]: sum h, sum h*x*/, int stat_ld, DwGeom g) {
  constexpr int P = (K - 1) / 2;
  constexpr int KK = K * K;
  constexpr int RELMIN = fdiv(-P, S);
  constexpr int RELMAX = fdiv(TW - 1 + P, S);
  constexpr int DW = RELMAX - RELMIN + 1;
  constexpr int MAXCH = 128;  // cvb*CV <= MAXCH (host guarantees)
  __shared__ float s_w[KK * MAXCH];
  __shared__ float s_red[MAXCH * (KK + 2)];

  const int tid = threadIdx.x;
  const int cvl = tid % g.cvb;
  const int pl = tid / g.cvb;
  const int cb0 = blockIdx.y * g.cvb * CV;  // first channel of this block
  const int c0 = cb0 + cvl * CV;
  const int nch = g.cvb * CV;
  const long gp = (long)blockIdx.x * g.pb + pl;
  const long nstrips = (long)g.N * g.nchunks * g.spr;
  const bool active = (pl < g.pb) && (c0 < g.C) && (gp < nstrips);

  for (int i = tid; i < KK * nch; i += 256) {
    const int t = i / nch, c = i % nch;
    s_w[t * nch + c] = (cb0 + c < g.C) ? w[(long)t * ldw + cb0 + c] : 0.f;
  }
  for (int i = tid; i < nch * (KK + 2); i += 256) s_red[i] = 0.f;
  __syncthreads();

  float dwa[KK][CV];
  float s0[CV], s1[CV];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int c = 0; c < CV; ++c) dwa[t][c] = 0.f;
#pragma unroll
  for (int c = 0; c < CV; ++c) s0[c] = s1[c] = 0.f;

  if (active) {
    const int ws = (int)(gp % g.spr);
    const int chunk = (int)((gp / g.spr) % g.nchunks);
    const int n = (int)(gp / ((long)g.spr * g.nchunks));
    const int wi0 = ws * TW;            // multiple of TW (TW % S == 0)
    const int wob = wi0 / S + RELMIN;   // first output column of the dY window

    float sc[CV], sh[CV], k1[CV], k2[CV], k3[CV];
    bool cvalid[CV];
#pragma unroll
    for (int c = 0; c < CV; ++c) { sc[c] = 1.f; sh[c] = 0.f; k1[c] = 1.f; k2[c] = 0.f; k3[c] = 0.f; cvalid[c] = (c0 + c) < g.C; }
    if (in_scale) { VecIO<float, CV>::load(in_scale + c0, sc); VecIO<float, CV>::load(in_shift + c0, sh); }
    if (c1) { VecIO<float, CV>::load(c1 + c0, k1); VecIO<float, CV>::load(c2 + c0, k2); VecIO<float, CV>::load(c3 + c0, k3); }

    const int hi_beg = chunk * g.rh;
    const int hi_end = min(g.H, hi_beg + g.rh);
    const T* xn = x + (long)n * g.H * g.W * ldx + c0;
    T* hn = h + (long)n * g.H * g.W * ldh + c0;
    const T* gn = gup + (long)n * g.Ho * g.Wo * ldg + c0;
    const T* yn = yraw ? yraw + (long)n * g.Ho * g.Wo * ldyr + c0 : nullptr;

    for (int hi = hi_beg; hi < hi_end; ++hi) {
      // this row's input pixels: raw value, activated value (for dW), relu mask
      float xraw[TW][CV], xa[TW][CV];
      bool pvalid[TW];
      const T* xr = xn + (long)hi * g.W * ldx;
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const int wi = wi0 + t;
        pvalid[t] = wi < g.W;
        if (pvalid[t]) {
          VecIO<T, CV>::load(xr + (long)wi * ldx, xraw[t]);
#pragma unroll
          for (int c = 0; c < CV; ++c) {
            float a = xraw[t][c] * sc[c] + sh[c];
            xa[t][c] = in_relu ? fmaxf(a, 0.f) : a;
          }
        } else {
#pragma unroll
          for (int c = 0; c < CV; ++c) { xraw[t][c] = 0.f; xa[t][c] = 0.f; }
        }
      }
      float dx[TW][CV];
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int c = 0; c < CV; ++c) dx[t][c] = 0.f;

#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int numr = hi + P - ky;
        if (numr < 0) continue;
        if (S > 1 && (numr % S) != 0) continue;
        const int ho = numr / S;
        if (ho >= g.Ho) continue;
        // dY window for this output row
        float dy[DW][CV];
        const T* gr = gn + (long)ho * g.Wo * ldg;
        const T* yr = yn ? yn + (long)ho * g.Wo * ldyr : nullptr;
#pragma unroll
        for (int j = 0; j < DW; ++j) {
          const int wo = wob + j;
          if (wo >= 0 && wo < g.Wo) {
            float gv[CV];
            VecIO<T, CV>::load(gr + (long)wo * ldg, gv);
            if (yr) {
              float yv[CV];
              VecIO<T, CV>::load(yr + (long)wo * ldyr, yv);
#pragma unroll
              for (int c = 0; c < CV; ++c) dy[j][c] = k1[c] * gv[c] + k2[c] * yv[c] + k3[c];
            } else {
#pragma unroll
              for (int c = 0; c < CV; ++c) dy[j][c] = k1[c] * gv[c];
            }
          } else {
#pragma unroll
            for (int c = 0; c < CV; ++c) dy[j][c] = 0.f;
          }
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          float wv[CV];
#pragma unroll
          for (int c = 0; c < CV; ++c) wv[c] = s_w[(ky * K + kx) * nch + cvl * CV + c];
#pragma unroll
          for (int t = 0; t < TW; ++t) {
            const int num = t + P - kx;                 // compile-time after unrolling
            if (pmod(num, S) != 0) continue;
            const int j = fdiv(num, S) - RELMIN;
#pragma unroll
            for (int c = 0; c < CV; ++c) {
              dx[t][c] += dy[j][c] * wv[c];
              dwa[ky * K + kx][c] += xa[t][c] * dy[j][c];
            }
          }
        }
      }

      T* hr = hn + (long)hi * g.W * ldh;
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        if (pvalid[t]) {
          float o[CV];
#pragma unroll
          for (int c = 0; c < CV; ++c) {
            float a = xraw[t][c] * sc[c] + sh[c];
            float v = (in_relu && !(a > 0.f)) ? 0.f : dx[t][c];
            v = cvalid[c] ? to_f32(from_f32<T>(v)) : 0.f;
            o[c] = v;
            s0[c] += v;
            s1[c] += v * xraw[t][c];
          }
          VecIO<T, CV>::store(hr + (long)(wi0 + t) * ldh, o);
        }
      }
    }
  }

  if (active) {
#pragma unroll
    for (int c = 0; c < CV; ++c) {
      float* r = &s_red[(cvl * CV + c) * (KK + 2)];
#pragma unroll
      for (int t = 0; t < KK; ++t) atomicAdd(&r[t], dwa[t][c]);
      atomicAdd(&r[KK], s0[c]);
      atomicAdd(&r[KK + 1], s1[c]);
    }
  }
  __syncthreads();
  for (int i = tid; i < nch * (KK + 2); i += 256) {
    const int cl = i / (KK + 2), t = i % (KK + 2);
    const int c = cb0 + cl;
    if (c >= g.C) continue;
    const float v = s_red[i];
    if (t < KK) {
      if (dw) atomicAdd(&dw[(long)c * KK + t], v);
    } else if (stats) {
      atomicAdd(&stats[(long)(t - KK) * stat_ld + c], v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static DwGeom make_geom(int N, int H, int W, int C, int Ho, int Wo, int CV, int TW, int rows, int cols, int max_cvb) {
  DwGeom g;
  g.N = N; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo;
  const int cvecs = (C + CV - 1) / CV;
  g.cvb = cvecs < max_cvb ? cvecs : max_cvb;
  g.pb = 256 / g.cvb;
  g.spr = (cols + TW - 1) / TW;
  // aim for >= ~4096 thread strips per channel group so that the chip is filled; otherwise split rows
  long strips = (long)N * g.spr;
  int nch = 1;
  while (strips * nch * ((cvecs + g.cvb - 1) / g.cvb) * g.cvb < 256L * 2048 && nch * 8 <= rows) nch *= 2;
  g.rh = (rows + nch - 1) / nch;
  if (g.rh % 2) g.rh += 1;  // keep chunk starts even (stride-2 row parity is then uniform across a wave)
  g.nchunks = (rows + g.rh - 1) / g.rh;
  return g;
}

template <typename T, int K, int S>
static int launch_fwd(const void* x, int ldx, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y,
                      int ldy, float* stats, int stat_ld, int N, int H, int W, int C, hipStream_t st) {
  constexpr int CV = 2, TW = 4;
  const int P = (K - 1) / 2;
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  DwGeom g = make_geom(N, H, W, C, Ho, Wo, CV, TW, Ho, Wo, 64);
  const int cvecs = (C + CV - 1) / CV;
  dim3 grid((unsigned)(((long)N * g.nchunks * g.spr + g.pb - 1) / g.pb), (cvecs + g.cvb - 1) / g.cvb);
  hipLaunchKernelGGL((k_dwconv_fwd<T, K, S, CV, TW>), grid, dim3(256), 0, st, (const T*)x, ldx, sc, sh, relu, w, ldw, (T*)y, ldy,
                     stats, stat_ld, g);
  return check_launch("dwconv_fwd");
}

template <typename T, int K, int S>
static int launch_bwd(const void* gup, int ldg, const void* yraw, int ldyr, const float* c1, const float* c2, const float* c3,
                      const void* x, int ldx, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h,
                      int ldh, float* dw, float* stats, int stat_ld, int N, int H, int W, int C, hipStream_t st) {
  constexpr int CV = 2, TW = 4;
  const int P = (K - 1) / 2;
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  DwGeom g = make_geom(N, H, W, C, Ho, Wo, CV, TW, H, W, 64);
  const int cvecs = (C + CV - 1) / CV;
  dim3 grid((unsigned)(((long)N * g.nchunks * g.spr + g.pb - 1) / g.pb), (cvecs + g.cvb - 1) / g.cvb);
  hipLaunchKernelGGL((k_dwconv_bwd<T, K, S, CV, TW>), grid, dim3(256), 0, st, (const T*)gup, ldg, (const T*)yraw, ldyr, c1, c2,
                     c3, (const T*)x, ldx, sc, sh, relu, w, ldw, (T*)h, ldh, dw, stats, stat_ld, g);
  return check_launch("dwconv_bwd");
}

#define DW_DISPATCH(FN, ...)                                                                   \
  do {                                                                                         \
    if (dtype == DT_F32) {                                                                     \
      if (k == 3 && stride == 1) return FN<float, 3, 1>(__VA_ARGS__);                          \
      if (k == 3 && stride == 2) return FN<float, 3, 2>(__VA_ARGS__);                          \
      if (k == 5 && stride == 1) return FN<float, 5, 1>(__VA_ARGS__);                          \
      if (k == 5 && stride == 2) return FN<float, 5, 2>(__VA_ARGS__);                          \
      if (k == 7 && stride == 1) return FN<float, 7, 1>(__VA_ARGS__);                          \
      if (k == 7 && stride == 2) return FN<float, 7, 2>(__VA_ARGS__);                          \
    } else {                                                                                   \
      if (k == 3 && stride == 1) return FN<bf16_t, 3, 1>(__VA_ARGS__);                         \
      if (k == 3 && stride == 2) return FN<bf16_t, 3, 2>(__VA_ARGS__);                         \
      if (k == 5 && stride == 1) return FN<bf16_t, 5, 1>(__VA_ARGS__);                         \
      if (k == 5 && stride == 2) return FN<bf16_t, 5, 2>(__VA_ARGS__);                         \
      if (k == 7 && stride == 1) return FN<bf16_t, 7, 1>(__VA_ARGS__);                         \
      if (k == 7 && stride == 2) return FN<bf16_t, 7, 2>(__VA_ARGS__);                         \
    }                                                                                          \
  } while (0)

}  // namespace atomnas

using namespace atomnas;

extern "C" int atomnas_dwconv_fwd(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_relu,
                                  const float* w, int ldw, void* y, int ldy, float* stats, int stat_ld, int N, int H, int W, int C,
                                  int k, int stride, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && w && y, "dwconv_fwd: null pointer");
  ATOMNAS_REQUIRE((k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2), "dwconv_fwd: unsupported k=%d stride=%d", k, stride);
  ATOMNAS_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv_fwd: bad dtype %d", dtype);
  ATOMNAS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "dwconv_fwd: empty shape");
  ATOMNAS_REQUIRE(ldx >= C && ldy >= C && ldw >= C && ldx % 2 == 0 && ldy % 2 == 0 && ldw % 2 == 0, "dwconv_fwd: bad pitch");
  ATOMNAS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dwconv_fwd: scale/shift must come together");
  hipStream_t st = (hipStream_t)stream;
  ATOMNAS_REQUIRE(!stats || stat_ld >= C, "dwconv_fwd: statistics pitch %d < C=%d", stat_ld, C);
  DW_DISPATCH(launch_fwd, x, ldx, in_scale, in_shift, in_relu, w, ldw, y, ldy, stats, stat_ld, N, H, W, C, st);
  return 1;
}

extern "C" int atomnas_dwconv_bwd(const void* g, int ldg, const void* yraw, int ldyr, const float* c1, const float* c2,
                                  const float* c3, const void* x, int ldx, const float* in_scale, const float* in_shift,
                                  int in_relu, const float* w, int ldw, void* h, int ldh, float* dw, float* stats, int stat_ld, int N,
                                  int H, int W, int C, int k, int stride, int dtype, void* stream) {
  ATOMNAS_REQUIRE(g && x && w && h, "dwconv_bwd: null pointer");
  ATOMNAS_REQUIRE((k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2), "dwconv_bwd: unsupported k=%d stride=%d", k, stride);
  ATOMNAS_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv_bwd: bad dtype %d", dtype);
  ATOMNAS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "dwconv_bwd: empty shape");
  ATOMNAS_REQUIRE(ldx >= C && ldg >= C && ldh >= C && ldw >= C && ldx % 2 == 0 && ldg % 2 == 0 && ldh % 2 == 0 && ldw % 2 == 0,
                  "dwconv_bwd: bad pitch");
  ATOMNAS_REQUIRE(!yraw || (c1 && c2 && c3 && ldyr >= C && ldyr % 2 == 0), "dwconv_bwd: yraw needs c1,c2,c3 and a valid pitch");
  ATOMNAS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dwconv_bwd: scale/shift must come together");
  hipStream_t st = (hipStream_t)stream;
  ATOMNAS_REQUIRE(!stats || stat_ld >= C, "dwconv_bwd: statistics pitch %d < C=%d", stat_ld, C);
  DW_DISPATCH(launch_bwd, g, ldg, yraw, ldyr, c1, c2, c3, x, ldx, in_scale, in_shift, in_relu, w, ldw, h, ldh, dw, stats, stat_ld,
              N, H, W, C, st);
  return 1;
}

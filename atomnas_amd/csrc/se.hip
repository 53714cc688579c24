// Squeeze-and-Excitation of the fused atomic block (AtomNAS+), forward and backward, on gfx950.
//
// Replaces the ATen calls behind SqueezeAndExcitation.forward (models/mobilenet_base.py:109-112) inside
// InvertedResidualChannelsFused.forward (:256-267):
//     s = mean_hw(A);  gate = sigmoid(W2 * act(W1 * s + b1) + b2);  S = gate * A          A = act(bn(D)), D = raw depthwise output
// The activated depthwise output A is never materialised: the squeeze, the gating and both backward passes re-derive it from
// the raw depthwise output D and the BatchNorm coefficients on load, like every other kernel of the block.  The hidden tensors
// (D, S, dS, g) are slab-major (include/atomnas_hip.h); the gate-sized tensors [N][HT] are plain fp32.
// The block lays its branches out in padded segments (HT channels); `cmap[c]` maps a padded channel to the row / column of
// the reference's SE weights ([hid][total], [total][hid]) or -1 for padding.
// Everything is reduced in a fixed order (per-image loops, then loops over the batch): no atomics.
#include "common.h"

namespace atomnas {

__device__ __forceinline__ float sigmoidf_(float a) { return 1.f / (1.f + __expf(-a)); }

// Per-image channel reductions over the pixels of one image (the squeeze and the gate gradient):
//   DG = false: out[n][c] = mean_hw act(D*scale+shift)                       (SqueezeAndExcitation.forward, mobilenet_base.py:110)
//   DG = true : out[n][c] = sum_hw dS[m][c] * act(D[m][c]*scale+shift)       (its backward through the gating product, :112)
// A workgroup owns one image and CGB groups of 8 channels; its 256 threads are CGB channel groups x PL = 256 / CGB pixel lanes,
// ordered (slab half, pixel lane, slab) so that on slab-major tensors a wave reads runs of PL x 32 contiguous bytes.  Every thread
// walks the pixels p = pl, pl + PL, ... with four independent 16-byte loads in flight; the PL partial sums of a channel are added in
// lane order by one thread (fixed order, no atomics).  Round 3 had one thread per (image, 8 channels) walking all HW pixels
// serially: 768 threads for the 48-channel 112 x 112 layer of AtomNAS-C+, 0.9 ms per launch.
template <typename T, bool DG>
__global__ __launch_bounds__(256) void k_se_pool(const T* __restrict__ d, int ldd, long dss, const T* __restrict__ ds, int ldds, long dsss,
                                                 const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                 float* __restrict__ out, int ldo, long pstride, int HW, int C, int cgb) {
  __shared__ float s_red[256 * 8];
  const int tid = threadIdx.x, n = blockIdx.x;
  // blockIdx.z: this workgroup's share of the image's pixels; its sums go to plane blockIdx.z of `out` (the consumer adds the planes)
  const int pchunk = (HW + gridDim.z - 1) / gridDim.z;
  const int p_lo = blockIdx.z * pchunk, p_hi = min(HW, p_lo + pchunk);
  out += (long)blockIdx.z * pstride;
  const int PL = 256 / cgb;
  int cgl, pl;
  if (cgb >= 2) { cgl = (tid & 1) + 2 * (tid / (2 * PL)); pl = (tid >> 1) % PL; } else { cgl = 0; pl = tid; }
  const int cg = blockIdx.y * cgb + cgl;
  const int c0 = cg * 8;
  const int ncg = (C + 7) / 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (cg < ncg) {
    float s[8], h[8];
    VecIO<float, 8>::load(scale + c0, s);
    VecIO<float, 8>::load(shift + c0, h);
    const Act m = act_of(act);
    const long row0 = (long)n * HW;
    int p = p_lo + pl;
    for (; p + 3 * PL < p_hi; p += 4 * PL) {
      float v[4][8], g[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        VecIO<T, 8>::load(d + lay_off(row0 + p + u * PL, c0, ldd, dss), v[u]);
        if (DG) VecIO<T, 8>::load(ds + lay_off(row0 + p + u * PL, c0, ldds, dsss), g[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = act_apply(v[u][e] * s[e] + h[e], m);
          acc[e] += DG ? g[u][e] * a : a;
        }
    }
    for (; p < p_hi; p += PL) {
      float v[8], g[8];
      VecIO<T, 8>::load(d + lay_off(row0 + p, c0, ldd, dss), v);
      if (DG) VecIO<T, 8>::load(ds + lay_off(row0 + p, c0, ldds, dsss), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = act_apply(v[e] * s[e] + h[e], m);
        acc[e] += DG ? g[e] * a : a;
      }
    }
  }
  // s_red[pl][cgl * 8 + e]
#pragma unroll
  for (int e = 0; e < 8; ++e) s_red[pl * (cgb * 8) + cgl * 8 + e] = acc[e];
  __syncthreads();
  if (tid < cgb * 8) {
    const int c = blockIdx.y * cgb * 8 + tid;
    if (c < ncg * 8) {
      float a = 0.f;
      for (int q = 0; q < PL; ++q) a += s_red[q * (cgb * 8) + tid];
      if (!DG) a *= 1.0f / (float)HW;
      out[(long)n * ldo + c] = (c < C) ? a : 0.f;
    }
  }
}

static int se_pool_cgb(int C) {
  const int ncg = (C + 7) / 8;
  int cgb = 2;
  while (cgb < ncg && cgb < 32) cgb *= 2;
  return cgb;
}

// Pixel shares per image of k_se_pool (planes of its output): enough workgroups to fill the chip on the early, few-channel layers
// (one workgroup per image left 128 workgroups streaming 25 MB: 108 us on the 112 x 112 layer of AtomNAS-C+), at least four
// 16-byte loads per thread and share.  A power of two, at most 16.
static int se_pool_parts(int N, int HW, int C) {
  const int cgb = se_pool_cgb(C), PL = 256 / cgb;
  const long ygrid = ((C + 7) / 8 + cgb - 1) / cgb;
  int parts = 1;
  while (parts < 16 && (long)N * ygrid * parts < 1024 && HW / (2 * parts) >= 4 * PL) parts *= 2;
  return parts;
}

// The two dense layers of the gate and their backward, for G images per workgroup (SqueezeAndExcitation.forward,
// models/mobilenet_base.py:110-112).  The weights arrive PACKED over the block's padded channel layout (runtime.py pack jobs, fp32):
//   wa [hid][HT]  rows of the layer whose reduction runs over the channels   (forward: W1;  backward: W2 transposed)
//   wb [hid][HT]  rows of the layer whose reduction runs over the hidden units (forward: W2 transposed;  backward: W1)
// so that every weight load is lane-contiguous and nothing is gathered through cmap; padding columns hold zeros.
//   forward  (BWD = false): hpre[n][j] = b1[j] + sum_c wa[j][c] * pooled[n][c];   gate[n][c] = sigmoid(b2p[c] + sum_j wb[j][c] * act(hpre[n][j]))
//   backward (BWD = true ): dz2[n][c] = dgate * gate * (1 - gate);  dz1[n][j] = act'(hpre[n][j]) * sum_c wa[j][c] * dz2[n][c];
//                           dpooled[n][c] = sum_j wb[j][c] * dz1[n][j]
// Phase 1: one wave per hidden unit, lanes over the channels, G accumulators per lane (one weight load feeds G images), butterfly
// sums.  Phase 2: one thread per channel, the hidden units in sequence, G accumulators.  All sums in a fixed order.
// Round 3 ran one workgroup of 256 threads per image with the weights gathered through cmap: every image re-read both weight
// matrices (128 x 1.8 MB for the last block of AtomNAS-C+) with uncoalesced rows in phase 2: 0.45 ms per launch there.
template <int G, bool BWD>
__global__ __launch_bounds__(1024) void k_se_mlp(const float* __restrict__ in, int parts, long pstride, float* __restrict__ in_sum,
                                                 const float* __restrict__ gate_in, int ldp, const int* __restrict__ cmap,
                                                 const float* __restrict__ wa, const float* __restrict__ ba, const float* __restrict__ wb,
                                                 const float* __restrict__ bb, int act, float* __restrict__ hio, float* __restrict__ mid,
                                                 float* __restrict__ out, float* __restrict__ dz2, int N, int HT, int hid) {
  extern __shared__ float s_se[];
  float* s_in = s_se;                        // [G][HT]: pooled (forward) / dz2 (backward)
  float* s_h = s_se + (size_t)G * HT;        // [hid][G]: act(hpre) (forward) / dz1 (backward)
  float* s_part = s_h + (size_t)hid * G;     // [JP][G][CW]: phase-2 partial sums when the hidden units are split over thread groups
  const int n0 = blockIdx.x * G, tid = threadIdx.x, NT = blockDim.x;
  const Act m = act_of(act);
  // Every global load below is unconditional (clamped index + select): loads behind a branch are waited for one by one.
  // Images past the batch are clamped to the last one: they recompute (and re-store) its values.
  // (uniform trip count, clamped element index: the loop unrolls with the loads of four elements in flight)
  const int tot = G * HT, nit = (tot + NT - 1) / NT;
  if (parts == 1) {
#pragma unroll 4
    for (int it = 0; it < nit; ++it) {
      const int i = min(tid + it * NT, tot - 1);
      const int g = i / HT, c = i - g * HT;
      const long o = (long)min(n0 + g, N - 1) * ldp + c;
      float v = in[o];
      if (BWD) {
        const float gt = gate_in[o];
        const int cm = cmap[c];
        v = v * gt * (1.f - gt);
        v = (cm >= 0) ? v : 0.f;
        dz2[o] = v;
      } else {
        in_sum[o] = v;
      }
      s_in[i] = (n0 + g < N) ? v : 0.f;
    }
  } else {   // the planes of k_se_pool, added in plane order (early layers: one or two elements per thread)
    for (int it = 0; it < nit; ++it) {
      const int i = min(tid + it * NT, tot - 1);
      const int g = i / HT, c = i - g * HT;
      const long o = (long)min(n0 + g, N - 1) * ldp + c;
      const float gt = BWD ? gate_in[o] : 0.f;
      const int cm = BWD ? cmap[c] : 0;
      float v = 0.f;
      for (int q0 = 0; q0 < parts; q0 += 4) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = in[(long)min(q0 + u, parts - 1) * pstride + o];
#pragma unroll
        for (int u = 0; u < 4; ++u) v += (q0 + u < parts) ? t[u] : 0.f;
      }
      if (BWD) {
        v = v * gt * (1.f - gt);
        v = (cm >= 0) ? v : 0.f;
        dz2[o] = v;
      } else {
        in_sum[o] = v;
      }
      s_in[i] = (n0 + g < N) ? v : 0.f;
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
  const int nci = (HT + 63) >> 6;
  for (int j = 2 * wave; j < hid; j += 2 * nw) {   // two hidden units per wave and pass: 2 x 4 weight loads in flight
    const bool two = j + 1 < hid;
    const float* w0 = wa + (long)j * HT;
    const float* w1 = wa + (long)(two ? j + 1 : j) * HT;
    float a0[G], a1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a0[g] = a1[g] = 0.f;
#pragma unroll 4
    for (int i = 0; i < nci; ++i) {
      const int c = lane + 64 * i, cl = min(c, HT - 1);
      float x0 = w0[cl], x1 = w1[cl];
      if (c >= HT) x0 = x1 = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float sv = s_in[g * HT + cl];
        a0[g] += x0 * sv;
        a1[g] += x1 * sv;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      a0[g] = wave_sum(a0[g]);
      a1[g] = wave_sum(a1[g]);
    }
    if (lane < 2 && (lane == 0 || two)) {
      const int jj = j + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float av = lane ? a1[g] : a0[g];
        const long o = (long)min(n0 + g, N - 1) * hid + jj;
        const bool live = n0 + g < N;
        float v;
        if (BWD) {
          v = act_bwd(av, hio[o], m);
          if (live) mid[o] = v;
        } else {
          const float pre = av + ba[jj];
          if (live) hio[o] = pre;
          v = act_apply(pre, m);
        }
        s_h[jj * G + g] = live ? v : 0.f;
      }
    }
  }
  __syncthreads();
  const int CW = min(NT, (HT + 63) & ~63), JP = NT / CW;
  if (JP == 1) {
    for (int c = tid; c < HT; c += NT) {
      float a[G];
      const float b0 = BWD ? 0.f : bb[c];
#pragma unroll
      for (int g = 0; g < G; ++g) a[g] = b0;
#pragma unroll 16
      for (int j = 0; j < hid; ++j) {
        const float wv = wb[(long)j * HT + c];
#pragma unroll
        for (int g = 0; g < G; ++g) a[g] += wv * s_h[j * G + g];
      }
      const bool valid = BWD || cmap[c] >= 0;
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (n0 + g < N) out[(long)(n0 + g) * ldp + c] = BWD ? a[g] : (valid ? sigmoidf_(a[g]) : 0.f);
    }
  } else {
    // few channels: JP thread groups share the hidden units of a channel; their partial sums are added in group order
    const int cq = tid % CW, jp = tid / CW, cl = min(cq, HT - 1);
    const int JL = (hid + JP - 1) / JP, j_lo = min(hid, jp * JL), j_hi = min(hid, j_lo + JL);
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a[g] = 0.f;
    if (jp < JP) {
#pragma unroll 8
      for (int j = j_lo; j < j_hi; ++j) {
        const float wv = wb[(long)j * HT + cl];
#pragma unroll
        for (int g = 0; g < G; ++g) a[g] += wv * s_h[j * G + g];
      }
#pragma unroll
      for (int g = 0; g < G; ++g) s_part[(jp * G + g) * CW + cq] = a[g];
    }
    __syncthreads();
    if (jp < JP && cq < HT) {
      const float b0 = BWD ? 0.f : bb[cq];
      const bool valid = BWD || cmap[cq] >= 0;
      for (int g = jp; g < G; g += JP) {
        float v = b0;
        for (int q = 0; q < JP; ++q) v += s_part[(q * G + g) * CW + cq];
        if (n0 + g < N) out[(long)(n0 + g) * ldp + cq] = BWD ? v : (valid ? sigmoidf_(v) : 0.f);
      }
    }
  }
}

template <bool BWD>
static int launch_se_mlp(const float* in, int parts, long pstride, float* in_sum, const float* gate_in, int ldp, const int* cmap,
                         const float* wa, const float* ba, const float* wb, const float* bb, int act, float* hio, float* mid, float* out,
                         float* dz2, int N, int HT, int hid, hipStream_t st) {
  const int NT = 1024;
  const size_t per_g = (size_t)(HT + hid + (HT <= NT / 2 ? NT : 0)) * sizeof(float);
#define ATOMNAS_SE_MLP(G_)                                                                                                             \
  do {                                                                                                                                 \
    const size_t lds = per_g * (G_);                                                                                                   \
    if (lds > 64 * 1024) (void)resident_per_cu(k_se_mlp<G_, BWD>, NT, lds);                                                            \
    hipLaunchKernelGGL((k_se_mlp<G_, BWD>), dim3((N + (G_) - 1) / (G_)), dim3(NT), lds, st, in, parts, pstride, in_sum, gate_in, ldp, cmap, \
                       wa, ba, wb, bb, act, hio, mid, out, dz2, N, HT, hid);                                                           \
    return 0;                                                                                                                          \
  } while (0)
  const size_t cap = max_lds_bytes();
  if (per_g * 8 <= cap) ATOMNAS_SE_MLP(8);
  if (per_g * 4 <= cap) ATOMNAS_SE_MLP(4);
  if (per_g * 2 <= cap) ATOMNAS_SE_MLP(2);
  if (per_g <= cap) ATOMNAS_SE_MLP(1);
#undef ATOMNAS_SE_MLP
  set_error("se_mlp: %d channels + %d hidden units do not fit the LDS", HT, hid);
  return 1;
}

// S = act(D*scale+shift) * gate[n]   (mode 0)        -- the projection's input
// dgate[n][c] = sum_hw dS * act(D*scale+shift)  is k_se_pool<T, true> above
template <typename T>
__global__ __launch_bounds__(256) void k_se_scale(const T* __restrict__ d, int ldd, long dss, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, int act, const float* __restrict__ gate, int ldg,
                                                  T* __restrict__ out, int ldo, long oss, long M, int HW, int C) {
  const int cg = (C + 7) / 8;
  const long total = M * cg;
  const Act m = act_of(act);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / cg;
    const int c0 = (int)(i % cg) * 8;
    const long n = row / HW;
    float v[8], s[8], h[8], gt[8];
    VecIO<T, 8>::load(d + lay_off(row, c0, ldd, dss), v);
    VecIO<float, 8>::load(scale + c0, s);
    VecIO<float, 8>::load(shift + c0, h);
    VecIO<float, 8>::load(gate + n * ldg + c0, gt);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? act_apply(v[e] * s[e] + h[e], m) * gt[e] : 0.f;
    VecIO<T, 8>::store(out + lay_off(row, c0, ldo, oss), v);
  }
}

// Weight gradients of the two dense layers, written through cmap into the reference's layouts (W1 [hid][total], W2 [total][hid]):
//   dW2[cc][j] += sum_n dz2[n][c] * act(hpre[n][j]);  db2[cc] += sum_n dz2[n][c];  dW1[j][cc] += sum_n dz1[n][j] * pooled[n][c];  db1[j] += sum_n dz1[n][j]
// A workgroup owns 64 channels x 32 hidden units of both products and walks the batch in chunks of 64 images staged in LDS
// (coalesced rows); a thread holds 4 channels x 2 hidden units of each product.  The batch is summed in image order.
// Round 3: one thread per output element with two dependent global loads per image, 50 us per launch whatever the size.
constexpr int SW_CT = 64, SW_JT = 32, SW_NB = 64;
__global__ __launch_bounds__(256) void k_se_wgrad(const float* __restrict__ dz2, const float* __restrict__ dz1, const float* __restrict__ pooled,
                                                  int ldg, const float* __restrict__ hpre, int act, const int* __restrict__ cmap,
                                                  float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                  float* __restrict__ db2, int N, int HT, int total, int hid) {
  __shared__ __attribute__((aligned(16))) float s_z2[SW_NB][SW_CT], s_p[SW_NB][SW_CT], s_ah[SW_NB][SW_JT], s_z1[SW_NB][SW_JT];
  const Act m = act_of(act);
  const int tid = threadIdx.x, c0 = blockIdx.x * SW_CT, j0 = blockIdx.y * SW_JT;
  const int cx = (tid & 15) * 4, jx = (tid >> 4) * 2;
  float a2[4][2], a1[4][2], sb2 = 0.f, sb1 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) a2[u][v] = a1[u][v] = 0.f;
  for (int nb = 0; nb < N; nb += SW_NB) {
    if (nb) __syncthreads();
    // unconditional loads at clamped indices, then selects: all 48 loads of a thread are in flight together (behind a branch
    // each was waited for on its own: 35 us per launch whatever the size)
#pragma unroll
    for (int it = 0; it < SW_NB * SW_CT / 256; ++it) {
      const int i = tid + it * 256;
      const int n = i / SW_CT, c = i % SW_CT;
      const bool ok = nb + n < N && c0 + c < HT;
      const long o = (long)min(nb + n, N - 1) * ldg + min(c0 + c, HT - 1);
      const float z = dz2[o], pv = pooled[o];
      s_z2[n][c] = ok ? z : 0.f;
      s_p[n][c] = ok ? pv : 0.f;
    }
    float hp[SW_NB * SW_JT / 256], zz[SW_NB * SW_JT / 256];
#pragma unroll
    for (int it = 0; it < SW_NB * SW_JT / 256; ++it) {
      const int i = tid + it * 256;
      const long o = (long)min(nb + i / SW_JT, N - 1) * hid + min(j0 + i % SW_JT, hid - 1);
      hp[it] = hpre[o];
      zz[it] = dz1[o];
    }
#pragma unroll
    for (int it = 0; it < SW_NB * SW_JT / 256; ++it) {   // the activation branches: kept apart from the loads
      const int i = tid + it * 256;
      const int n = i / SW_JT, j = i % SW_JT;
      const bool ok = nb + n < N && j0 + j < hid;
      s_ah[n][j] = ok ? act_apply(hp[it], m) : 0.f;
      s_z1[n][j] = ok ? zz[it] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int n = 0; n < SW_NB; ++n) {
      const f32x4 z2 = *reinterpret_cast<const f32x4*>(&s_z2[n][cx]);
      const f32x4 pp = *reinterpret_cast<const f32x4*>(&s_p[n][cx]);
      const float h0 = s_ah[n][jx], h1 = s_ah[n][jx + 1], z0 = s_z1[n][jx], z1 = s_z1[n][jx + 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a2[u][0] += z2[u] * h0;
        a2[u][1] += z2[u] * h1;
        a1[u][0] += z0 * pp[u];
        a1[u][1] += z1 * pp[u];
      }
    }
    // bias gradients: four interleaved partial sums per chunk, combined in a fixed order
    if (blockIdx.y == 0 && tid < SW_CT) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int n = 0; n < SW_NB; n += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] += s_z2[n + u][tid];
      sb2 += (t[0] + t[1]) + (t[2] + t[3]);
    }
    if (blockIdx.x == 0 && tid >= 64 && tid < 64 + SW_JT) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int n = 0; n < SW_NB; n += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] += s_z1[n + u][tid - 64];
      sb1 += (t[0] + t[1]) + (t[2] + t[3]);
    }
  }
  // accumulate into the gradient arena: all 16 reads first, unconditionally at clamped addresses, then the guarded writes
  int cc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) cc[u] = cmap[min(c0 + cx + u, HT - 1)];
  float o2[4][2], o1[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int ccl = max(cc[u], 0), jl = min(j0 + jx + v, hid - 1);
      o2[u][v] = dw2[(long)ccl * hid + jl];
      o1[u][v] = dw1[(long)jl * total + ccl];
    }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int j = j0 + jx + v;
      if (c0 + cx + u < HT && cc[u] >= 0 && j < hid) {
        dw2[(long)cc[u] * hid + j] = o2[u][v] + a2[u][v];
        dw1[(long)j * total + cc[u]] = o1[u][v] + a1[u][v];
      }
    }
  if (blockIdx.y == 0 && tid < SW_CT && c0 + tid < HT) {
    const int cb = cmap[c0 + tid];
    if (cb >= 0) db2[cb] += sb2;
  }
  if (blockIdx.x == 0 && tid >= 64 && tid < 64 + SW_JT && j0 + tid - 64 < hid) db1[j0 + tid - 64] += sb1;
}

// g = act'(a) * (dS * gate[n] + dpooled[n] / HW),  a = D*scale+shift;  stats2 rows = [sum g, sum g*D]
// Thread mapping of k_se_pool: a workgroup is cgb groups of 8 channels x PL = 256 / cgb pixel lanes, ordered (slab half, pixel lane,
// slab) so that a wave reads runs of PL x 32 contiguous bytes of a slab-major tensor; two pixels per thread in flight; the pixel
// lanes are combined in lane order.  (Round 3 fixed 32 channel groups x 8 pixel lanes: 19 % of the threads had channels on the
// 48-channel 112 x 112 layer of AtomNAS-C+, 0.48 ms per launch.)
template <typename T>
__global__ __launch_bounds__(256) void k_se_bwd_apply(const T* __restrict__ ds, int ldds, long dsss, const T* __restrict__ d, int ldd, long dss,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                      const float* __restrict__ gate, const float* __restrict__ dpooled, int ldg,
                                                      T* __restrict__ g, int ldgo, long gss, float* __restrict__ stats2, int stat_rows,
                                                      long M, int HW, int C, int cgb) {
  __shared__ float s_red[256 * 16];   // [PL][cgb * 8][2]
  const int tid = threadIdx.x;
  const int PL = 256 / cgb;
  const int cgl = (tid & 1) + 2 * (tid / (2 * PL)), pl = (tid >> 1) % PL;
  const int c0 = (blockIdx.y * cgb + cgl) * 8;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  if (c0 < C) {
    float s[8], h[8];
    VecIO<float, 8>::load(scale + c0, s);
    VecIO<float, 8>::load(shift + c0, h);
    const Act m = act_of(act);
    const float inv = 1.0f / (float)HW;
    const long step = (long)gridDim.x * PL;
    for (long p = (long)blockIdx.x * PL + pl; p < M; p += 2 * step) {
      float dv[2][8], v[2][8], gt[2][8], dp[2][8];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long q = p + u * step;
        ok[u] = q < M;
        if (ok[u]) {
          const long n = q / HW;
          VecIO<T, 8>::load(ds + lay_off(q, c0, ldds, dsss), dv[u]);
          VecIO<T, 8>::load(d + lay_off(q, c0, ldd, dss), v[u]);
          VecIO<float, 8>::load(gate + n * ldg + c0, gt[u]);
          VecIO<float, 8>::load(dpooled + n * ldg + c0, dp[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = v[u][e] * s[e] + h[e];
          float gg = act_bwd(dv[u][e] * gt[u][e] + dp[u][e] * inv, a, m);
          if (c0 + e >= C) gg = 0.f;
          gg = to_f32(from_f32<T>(gg));
          dv[u][e] = gg;
          s0[e] += gg;
          s1[e] += gg * v[u][e];
        }
        VecIO<T, 8>::store(g + lay_off(p + u * step, c0, ldgo, gss), dv[u]);
      }
    }
  }
  const int wch = cgb * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s_red[(pl * wch + cgl * 8 + e) * 2] = s0[e];
    s_red[(pl * wch + cgl * 8 + e) * 2 + 1] = s1[e];
  }
  __syncthreads();
  if (stats2 && tid < wch) {
    const int c = blockIdx.y * wch + tid;
    if (c < C) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < PL; ++q) {
        a += s_red[(q * wch + tid) * 2];
        b += s_red[(q * wch + tid) * 2 + 1];
      }
      float* srow = stats2 + (long)blockIdx.x * 2 * C;
      srow[c] = a;
      srow[C + c] = b;
      stat_zero_tail(stats2, 2L * C, blockIdx.x + gridDim.x, gridDim.x, stat_rows, c);
      stat_zero_tail(stats2, 2L * C, blockIdx.x + gridDim.x, gridDim.x, stat_rows, C + c);
    }
  }
}

}  // namespace atomnas

using namespace atomnas;

extern "C" int atomnas_se_pool_parts(int N, int HW, int C) { return (N > 0 && HW > 0 && C > 0) ? se_pool_parts(N, HW, C) : 1; }

extern "C" int atomnas_se_squeeze(const void* d, int ldd, long d_ss, const float* scale, const float* shift, int act, float* pooled,
                                  int ldp, int parts, long part_stride, int N, int HW, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(d && scale && shift && pooled && N > 0 && HW > 0 && C > 0 && parts >= 1 && parts <= 16, "se_squeeze: bad arguments");
  const int cgb = se_pool_cgb(C);
  const dim3 grid(N, ((C + 7) / 8 + cgb - 1) / cgb, parts);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL((k_se_pool<float, false>), grid, dim3(256), 0, st, (const float*)d, ldd, d_ss, (const float*)nullptr, 0, 0L, scale, shift,
                       act, pooled, ldp, part_stride, HW, C, cgb);
  else
    hipLaunchKernelGGL((k_se_pool<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)d, ldd, d_ss, (const bf16_t*)nullptr, 0, 0L, scale,
                       shift, act, pooled, ldp, part_stride, HW, C, cgb);
  return check_launch("se_squeeze");
}

extern "C" int atomnas_se_mlp_fwd(const float* pooled_parts, int ldp, int parts, long part_stride, float* pooled, const int* cmap,
                                  const float* w1p, const float* b1, const float* w2t, const float* b2p, int act, float* hpre, float* gate,
                                  int N, int HT, int hid, void* stream) {
  ATOMNAS_REQUIRE(pooled_parts && pooled && cmap && w1p && b1 && w2t && b2p && hpre && gate && N > 0 && HT > 0 && hid > 0 && parts >= 1,
                  "se_mlp_fwd: bad arguments");
  if (launch_se_mlp<false>(pooled_parts, parts, part_stride, pooled, nullptr, ldp, cmap, w1p, b1, w2t, b2p, act, hpre, nullptr, gate, nullptr,
                           N, HT, hid, (hipStream_t)stream))
    return 1;
  return check_launch("se_mlp_fwd");
}

extern "C" int atomnas_se_scale(const void* d, int ldd, long d_ss, const float* scale, const float* shift, int act, const float* gate,
                                int ldg, void* out, int ldo, long o_ss, long M, int HW, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(d && scale && shift && gate && out && M > 0 && HW > 0 && C > 0, "se_scale: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  long blocks = (M * ((C + 7) / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_se_scale<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d, ldd, d_ss, scale, shift, act, gate, ldg,
                       (float*)out, ldo, o_ss, M, HW, C);
  else
    hipLaunchKernelGGL(k_se_scale<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)d, ldd, d_ss, scale, shift, act, gate,
                       ldg, (bf16_t*)out, ldo, o_ss, M, HW, C);
  return check_launch("se_scale");
}

// backward of the gate: dgate -> (per image) dz2, dz1, dpooled -> SE weight gradients (accumulated into dw1, db1, dw2, db2)
extern "C" int atomnas_se_bwd_gate(const void* ds, int ldds, long ds_ss, const void* d, int ldd, long d_ss, const float* scale,
                                   const float* shift, int act, const float* gate, const float* pooled, int ldg, const int* cmap,
                                   const float* w1p, const float* w2t, const float* hpre, float* dgate, int parts, long part_stride,
                                   float* dz2, float* dz1, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, int se_act, int N, int HW, int HT,
                                   int total, int hid, int dtype, void* stream) {
  ATOMNAS_REQUIRE(ds && d && scale && shift && gate && pooled && cmap && w1p && w2t && hpre && dgate && dz2 && dz1 && dpooled && dw1 &&
                      db1 && dw2 && db2 && N > 0 && HW > 0 && HT > 0 && hid > 0,
                  "se_bwd_gate: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int cgb = se_pool_cgb(HT);
  ATOMNAS_REQUIRE(parts >= 1 && parts <= 16, "se_bwd_gate: dgate has 1..16 planes (atomnas_se_pool_parts)");
  const dim3 grid(N, ((HT + 7) / 8 + cgb - 1) / cgb, parts);
  if (dtype == DT_F32)
    hipLaunchKernelGGL((k_se_pool<float, true>), grid, dim3(256), 0, st, (const float*)d, ldd, d_ss, (const float*)ds, ldds, ds_ss, scale, shift,
                       act, dgate, ldg, part_stride, HW, HT, cgb);
  else
    hipLaunchKernelGGL((k_se_pool<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)d, ldd, d_ss, (const bf16_t*)ds, ldds, ds_ss, scale,
                       shift, act, dgate, ldg, part_stride, HW, HT, cgb);
  if (launch_se_mlp<true>(dgate, parts, part_stride, nullptr, gate, ldg, cmap, w2t, nullptr, w1p, nullptr, se_act, const_cast<float*>(hpre), dz1, dpooled, dz2, N, HT, hid, st))
    return 1;
  hipLaunchKernelGGL(k_se_wgrad, dim3((HT + SW_CT - 1) / SW_CT, (hid + SW_JT - 1) / SW_JT), dim3(256), 0, st, dz2, dz1, pooled, ldg, hpre, se_act,
                     cmap, dw1, db1, dw2, db2, N, HT, total, hid);
  return check_launch("se_bwd_gate");
}

extern "C" int atomnas_se_bwd_apply(const void* ds, int ldds, long ds_ss, const void* d, int ldd, long d_ss, const float* scale,
                                    const float* shift, int act, const float* gate, const float* dpooled, int ldg, void* g, int ldgo,
                                    long g_ss, float* stats2, int stat_rows, long M, int HW, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(ds && d && scale && shift && gate && dpooled && g && M > 0 && HW > 0 && C > 0, "se_bwd_apply: bad arguments");
  ATOMNAS_REQUIRE(!stats2 || stat_rows > 0, "se_bwd_apply: statistics need stat_rows > 0");
  const int cgb = se_pool_cgb(C), PL = 256 / cgb;
  long gx = (M + 2L * PL - 1) / (2L * PL);
  if (gx > 1024) gx = 1024;
  if (stats2 && gx > stat_rows) gx = stat_rows;
  dim3 grid((unsigned)gx, ((C + 7) / 8 + cgb - 1) / cgb);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_se_bwd_apply<float>, grid, dim3(256), 0, st, (const float*)ds, ldds, ds_ss, (const float*)d, ldd, d_ss, scale, shift,
                       act, gate, dpooled, ldg, (float*)g, ldgo, g_ss, stats2, stat_rows, M, HW, C, cgb);
  else
    hipLaunchKernelGGL(k_se_bwd_apply<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)ds, ldds, ds_ss, (const bf16_t*)d, ldd, d_ss, scale,
                       shift, act, gate, dpooled, ldg, (bf16_t*)g, ldgo, g_ss, stats2, stat_rows, M, HW, C, cgb);
  return check_launch("se_bwd_apply");
}

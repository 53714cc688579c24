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
                                                 float* __restrict__ out, int ldo, int HW, int C, int cgb) {
  __shared__ float s_red[256 * 8];
  const int tid = threadIdx.x, n = blockIdx.x;
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
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {
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
    for (; p < HW; p += PL) {
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

// one workgroup per image: hpre[n][j] = b1[j] + sum_c W1[j][cmap c] * pooled[n][c];  gate[n][c] = sigmoid(b2 + sum_j W2[cmap c][j] * act(hpre))
__global__ __launch_bounds__(256) void k_se_mlp_fwd(const float* __restrict__ pooled, int ldp, const int* __restrict__ cmap,
                                                    const float* __restrict__ w1, const float* __restrict__ b1,
                                                    const float* __restrict__ w2, const float* __restrict__ b2, int act,
                                                    float* __restrict__ hpre, float* __restrict__ gate, int HT, int total, int hid) {
  extern __shared__ float s_h[];   // [hid] activated hidden units
  const int n = blockIdx.x;
  const Act m = act_of(act);
  const float* pn = pooled + (long)n * ldp;
  // one wave per hidden unit, the lanes over the channels (rows of W1 are contiguous over the channels: coalesced), butterfly sum
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < hid; j += 4) {
    float a = 0.f;
    for (int c = lane; c < HT; c += 64) {
      const int cc = cmap[c];
      if (cc >= 0) a += w1[(long)j * total + cc] * pn[c];
    }
    a = wave_sum(a) + b1[j];
    if (lane == 0) {
      hpre[(long)n * hid + j] = a;
      s_h[j] = act_apply(a, m);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < HT; c += 256) {
    const int cc = cmap[c];
    float gt = 0.f;
    if (cc >= 0) {
      float a = b2[cc];
      for (int j = 0; j < hid; ++j) a += w2[(long)cc * hid + j] * s_h[j];
      gt = sigmoidf_(a);
    }
    gate[(long)n * ldp + c] = gt;
  }
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

// per image: dz2[n][c] = dgate * gate * (1 - gate);  dz1[n][j] = act'(hpre) * sum_c W2[cmap c][j] * dz2;  dpooled[n][c] = sum_j W1[j][cmap c] * dz1
__global__ __launch_bounds__(256) void k_se_mlp_bwd_img(const float* __restrict__ dgate, const float* __restrict__ gate, int ldg,
                                                        const int* __restrict__ cmap, const float* __restrict__ w1,
                                                        const float* __restrict__ w2, const float* __restrict__ hpre, int act,
                                                        float* __restrict__ dz2, float* __restrict__ dz1, float* __restrict__ dpooled,
                                                        int HT, int total, int hid) {
  extern __shared__ float s_buf[];   // [HT] dz2 of this image, then [hid] dz1
  float* s_z2 = s_buf;
  float* s_z1 = s_buf + HT;
  const int n = blockIdx.x;
  const Act m = act_of(act);
  for (int c = threadIdx.x; c < HT; c += 256) {
    const float gt = gate[(long)n * ldg + c];
    const float v = (cmap[c] >= 0) ? dgate[(long)n * ldg + c] * gt * (1.f - gt) : 0.f;
    s_z2[c] = v;
    dz2[(long)n * ldg + c] = v;
  }
  __syncthreads();
  // dz1: the 256 threads are JP hidden units x (256 / JP) channel ranges (rows of W2 are contiguous over the hidden units: coalesced);
  // the range partials are added in range order
  {
    int JP = 1;
    while (JP < hid && JP < 256) JP *= 2;
    const int parts = 256 / JP, j = threadIdx.x % JP, part = threadIdx.x / JP;
    float* s_part = s_z1 + hid;   // [parts][JP]
    for (int j0 = 0; j0 < hid; j0 += JP) {
      float a = 0.f;
      if (j0 + j < hid)
        for (int c = part; c < HT; c += parts) {
          const int cc = cmap[c];
          if (cc >= 0) a += w2[(long)cc * hid + j0 + j] * s_z2[c];
        }
      s_part[part * JP + j] = a;
      __syncthreads();
      if (part == 0 && j0 + j < hid) {
        float t = s_part[j];
        for (int q = 1; q < parts; ++q) t += s_part[q * JP + j];
        const float v = act_bwd(t, hpre[(long)n * hid + j0 + j], m);
        s_z1[j0 + j] = v;
        dz1[(long)n * hid + j0 + j] = v;
      }
      __syncthreads();
    }
  }
  for (int c = threadIdx.x; c < HT; c += 256) {
    const int cc = cmap[c];
    float a = 0.f;
    if (cc >= 0)
      for (int j = 0; j < hid; ++j) a += w1[(long)j * total + cc] * s_z1[j];
    dpooled[(long)n * ldg + c] = a;
  }
}

// weight gradients, batch loop in image order: one thread per output element
//   dW2[cc][j] += sum_n dz2[n][c] * act(hpre[n][j]);  db2[cc] += sum_n dz2[n][c];  dW1[j][cc] += sum_n dz1[n][j] * pooled[n][c];  db1[j] += sum_n dz1[n][j]
__global__ __launch_bounds__(256) void k_se_wgrad(const float* __restrict__ dz2, const float* __restrict__ dz1, const float* __restrict__ pooled,
                                                  int ldg, const float* __restrict__ hpre, int act, const int* __restrict__ cmap,
                                                  float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                  float* __restrict__ db2, int N, int HT, int total, int hid) {
  const Act m = act_of(act);
  const long n_w = (long)HT * hid;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 2 * n_w + HT + hid; i += (long)gridDim.x * 256) {
    if (i < n_w) {   // dW2
      const int c = (int)(i / hid), j = (int)(i % hid);
      const int cc = cmap[c];
      if (cc < 0) continue;
      float a = 0.f;
      for (int n = 0; n < N; ++n) a += dz2[(long)n * ldg + c] * act_apply(hpre[(long)n * hid + j], m);
      dw2[(long)cc * hid + j] += a;
    } else if (i < 2 * n_w) {   // dW1
      const long k = i - n_w;
      const int c = (int)(k / hid), j = (int)(k % hid);
      const int cc = cmap[c];
      if (cc < 0) continue;
      float a = 0.f;
      for (int n = 0; n < N; ++n) a += dz1[(long)n * hid + j] * pooled[(long)n * ldg + c];
      dw1[(long)j * total + cc] += a;
    } else if (i < 2 * n_w + HT) {   // db2
      const int c = (int)(i - 2 * n_w);
      const int cc = cmap[c];
      if (cc < 0) continue;
      float a = 0.f;
      for (int n = 0; n < N; ++n) a += dz2[(long)n * ldg + c];
      db2[cc] += a;
    } else {   // db1
      const int j = (int)(i - 2 * n_w - HT);
      float a = 0.f;
      for (int n = 0; n < N; ++n) a += dz1[(long)n * hid + j];
      db1[j] += a;
    }
  }
}

// g = act'(a) * (dS * gate[n] + dpooled[n] / HW),  a = D*scale+shift;  stats2 rows = [sum g, sum g*D]
// block = 32 channel groups x 8 pixel lanes (the hidden tensor is wide); pixel lanes combined in lane order
template <typename T>
__global__ __launch_bounds__(256) void k_se_bwd_apply(const T* __restrict__ ds, int ldds, long dsss, const T* __restrict__ d, int ldd, long dss,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                      const float* __restrict__ gate, const float* __restrict__ dpooled, int ldg,
                                                      T* __restrict__ g, int ldgo, long gss, float* __restrict__ stats2, int stat_rows,
                                                      long M, int HW, int C) {
  __shared__ float s_red[8][512];
  const int tid = threadIdx.x;
  const int cgl = tid & 31, pl = tid >> 5;
  const int c0 = (blockIdx.y * 32 + cgl) * 8;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  if (c0 < C) {
    float s[8], h[8];
    VecIO<float, 8>::load(scale + c0, s);
    VecIO<float, 8>::load(shift + c0, h);
    const Act m = act_of(act);
    const float inv = 1.0f / (float)HW;
    for (long p = (long)blockIdx.x * 8 + pl; p < M; p += (long)gridDim.x * 8) {
      const long n = p / HW;
      float dv[8], v[8], gt[8], dp[8];
      VecIO<T, 8>::load(ds + lay_off(p, c0, ldds, dsss), dv);
      VecIO<T, 8>::load(d + lay_off(p, c0, ldd, dss), v);
      VecIO<float, 8>::load(gate + n * ldg + c0, gt);
      VecIO<float, 8>::load(dpooled + n * ldg + c0, dp);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = v[e] * s[e] + h[e];
        float gg = act_bwd(dv[e] * gt[e] + dp[e] * inv, a, m);
        if (c0 + e >= C) gg = 0.f;
        gg = to_f32(from_f32<T>(gg));
        dv[e] = gg;
        s0[e] += gg;
        s1[e] += gg * v[e];
      }
      VecIO<T, 8>::store(g + lay_off(p, c0, ldgo, gss), dv);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s_red[pl][(cgl * 8 + e) * 2] = s0[e];
    s_red[pl][(cgl * 8 + e) * 2 + 1] = s1[e];
  }
  __syncthreads();
  if (stats2) {
    const int c = blockIdx.y * 256 + tid;
    if (c < C) {
      float a = s_red[0][tid * 2], b = s_red[0][tid * 2 + 1];
#pragma unroll
      for (int q = 1; q < 8; ++q) { a += s_red[q][tid * 2]; b += s_red[q][tid * 2 + 1]; }
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

extern "C" int atomnas_se_squeeze(const void* d, int ldd, long d_ss, const float* scale, const float* shift, int act, float* pooled,
                                  int ldp, int N, int HW, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(d && scale && shift && pooled && N > 0 && HW > 0 && C > 0 && ldp >= (C + 7) / 8 * 8, "se_squeeze: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int cgb = se_pool_cgb(C);
  const dim3 grid(N, ((C + 7) / 8 + cgb - 1) / cgb);
  if (dtype == DT_F32)
    hipLaunchKernelGGL((k_se_pool<float, false>), grid, dim3(256), 0, st, (const float*)d, ldd, d_ss, (const float*)nullptr, 0, 0L, scale, shift,
                       act, pooled, ldp, HW, C, cgb);
  else
    hipLaunchKernelGGL((k_se_pool<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)d, ldd, d_ss, (const bf16_t*)nullptr, 0, 0L, scale,
                       shift, act, pooled, ldp, HW, C, cgb);
  return check_launch("se_squeeze");
}

extern "C" int atomnas_se_mlp_fwd(const float* pooled, int ldp, const int* cmap, const float* w1, const float* b1, const float* w2,
                                  const float* b2, int act, float* hpre, float* gate, int N, int HT, int total, int hid, void* stream) {
  ATOMNAS_REQUIRE(pooled && cmap && w1 && b1 && w2 && b2 && hpre && gate && N > 0 && HT > 0 && total > 0 && hid > 0 && ldp >= HT,
                  "se_mlp_fwd: bad arguments");
  ATOMNAS_REQUIRE((size_t)hid * sizeof(float) <= 48 * 1024, "se_mlp_fwd: hidden width %d too large", hid);
  hipLaunchKernelGGL(k_se_mlp_fwd, dim3(N), dim3(256), (size_t)hid * sizeof(float), (hipStream_t)stream, pooled, ldp, cmap, w1, b1, w2, b2,
                     act, hpre, gate, HT, total, hid);
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
                                   const float* w1, const float* w2, const float* hpre, float* dgate, float* dz2, float* dz1,
                                   float* dpooled, float* dw1, float* db1, float* dw2, float* db2, int N, int HW, int HT, int total,
                                   int hid, int dtype, void* stream) {
  ATOMNAS_REQUIRE(ds && d && scale && shift && gate && pooled && cmap && w1 && w2 && hpre && dgate && dz2 && dz1 && dpooled && dw1 &&
                      db1 && dw2 && db2 && N > 0 && HW > 0 && HT > 0 && hid > 0,
                  "se_bwd_gate: bad arguments");
  ATOMNAS_REQUIRE((size_t)(HT + hid + 256) * sizeof(float) <= 60 * 1024, "se_bwd_gate: block too wide for the per-image kernel");
  hipStream_t st = (hipStream_t)stream;
  const int cgb = se_pool_cgb(HT);
  const dim3 grid(N, ((HT + 7) / 8 + cgb - 1) / cgb);
  if (dtype == DT_F32)
    hipLaunchKernelGGL((k_se_pool<float, true>), grid, dim3(256), 0, st, (const float*)d, ldd, d_ss, (const float*)ds, ldds, ds_ss, scale, shift,
                       act, dgate, ldg, HW, HT, cgb);
  else
    hipLaunchKernelGGL((k_se_pool<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)d, ldd, d_ss, (const bf16_t*)ds, ldds, ds_ss, scale,
                       shift, act, dgate, ldg, HW, HT, cgb);
  hipLaunchKernelGGL(k_se_mlp_bwd_img, dim3(N), dim3(256), (size_t)(HT + hid + 256) * sizeof(float), st, dgate, gate, ldg, cmap, w1, w2, hpre, act,
                     dz2, dz1, dpooled, HT, total, hid);
  long elems = 2L * HT * hid + HT + hid;
  long blocks = (elems + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_se_wgrad, dim3((unsigned)blocks), dim3(256), 0, st, dz2, dz1, pooled, ldg, hpre, act, cmap, dw1, db1, dw2, db2, N, HT,
                     total, hid);
  return check_launch("se_bwd_gate");
}

extern "C" int atomnas_se_bwd_apply(const void* ds, int ldds, long ds_ss, const void* d, int ldd, long d_ss, const float* scale,
                                    const float* shift, int act, const float* gate, const float* dpooled, int ldg, void* g, int ldgo,
                                    long g_ss, float* stats2, int stat_rows, long M, int HW, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(ds && d && scale && shift && gate && dpooled && g && M > 0 && HW > 0 && C > 0, "se_bwd_apply: bad arguments");
  ATOMNAS_REQUIRE(!stats2 || stat_rows > 0, "se_bwd_apply: statistics need stat_rows > 0");
  long gx = (M + 7) / 8;
  if (gx > 1024) gx = 1024;
  if (stats2 && gx > stat_rows) gx = stat_rows;
  dim3 grid((unsigned)gx, (C + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_se_bwd_apply<float>, grid, dim3(256), 0, st, (const float*)ds, ldds, ds_ss, (const float*)d, ldd, d_ss, scale, shift,
                       act, gate, dpooled, ldg, (float*)g, ldgo, g_ss, stats2, stat_rows, M, HW, C);
  else
    hipLaunchKernelGGL(k_se_bwd_apply<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)ds, ldds, ds_ss, (const bf16_t*)d, ldd, d_ss, scale,
                       shift, act, gate, dpooled, ldg, (bf16_t*)g, ldgo, g_ss, stats2, stat_rows, M, HW, C);
  return check_launch("se_bwd_apply");
}

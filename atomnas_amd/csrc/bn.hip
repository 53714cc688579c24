// BatchNorm pieces that are not fused into a convolution: the per-channel "finalize" steps between the producing
// kernel's statistics epilogue and the consuming kernel's apply prologue, and the few stand-alone element-wise passes.
//
// Semantics follow torch.nn.BatchNorm2d as instantiated by the reference (models/mobilenet_base.py:142,342 with
// momentum/eps from models/mobilenet_supernet.py:95-98): biased variance for normalisation, unbiased variance for the
// running estimate, running = (1-m)*running + m*batch, m = 1/num_batches_tracked in cumulative mode (momentum=None, used
// by utils/common.py:214-226 for calibration).  The backward finalize also adds the resource-weighted L1 sub-gradient
// rho*penalty*sign(gamma) of utils/prune.py:161-167 to dgamma.
#include "common.h"

namespace atomnas {

// Sums the partial rows of 16 channels in a fixed order.  The 256 threads of a block are 64 row groups x 4 channel quads:
// thread (rg, cq) adds rows rg, rg+64, ... of channels 4cq..4cq+3 (both planes) with 16-byte loads, all of them independent
// (one round of memory latency for up to 512 rows); the 64 group sums of a channel are then added in order 0..63 by one
// thread per (channel, plane).  Results: t0/t1 of channel c = blockIdx.x*16 + (tid & 15), valid in threads tid < 16.
__device__ __forceinline__ void stat_row_sum(const float* __restrict__ stats, int rows, int ld, int C, float (&s_red)[2][64][17], float& t0,
                                             float& t1) {
  const int tid = threadIdx.x;
  const int rg = tid >> 2, cq = tid & 3;
  const int c4 = blockIdx.x * 16 + cq * 4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  if (c4 < C) {   // C is a multiple of 4 here (channel vectors are padded to 8)
#pragma unroll 4
    for (int r = rg; r < rows; r += 64) {
      a += *reinterpret_cast<const f32x4*>(stats + (long)r * 2 * ld + c4);
      b += *reinterpret_cast<const f32x4*>(stats + (long)r * 2 * ld + ld + c4);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s_red[0][rg][cq * 4 + e] = a[e];
    s_red[1][rg][cq * 4 + e] = b[e];
  }
  __syncthreads();
  t0 = t1 = 0.f;
  if (tid < 32) {
    const int pl = tid >> 4, cl = tid & 15;
    float t = 0.f;
#pragma unroll 8
    for (int g = 0; g < 64; ++g) t += s_red[pl][g][cl];
    s_red[pl][0][cl] = t;
  }
  __syncthreads();
  if (tid < 16) { t0 = s_red[0][0][tid]; t1 = s_red[1][0][tid]; }
}

__global__ __launch_bounds__(256) void k_bn_finalize_fwd(const float* __restrict__ stats, int rows, int stat_ld, float inv_count, float unbias,
                                  const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                  float* __restrict__ running_var, long long* __restrict__ nbt, float* __restrict__ scale,
                                  float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd, int C,
                                  int Cpad, const int* __restrict__ cmap) {
  __shared__ float s_red[2][64][17];
  const int c = blockIdx.x * 16 + threadIdx.x;
  float a0, a1;
  stat_row_sum(stats, rows, stat_ld, Cpad, s_red, a0, a1);
  if (threadIdx.x >= 16 || c >= Cpad) return;
  // cmap (fused block: the kernels' padded branch segments against the module's contiguous parameter vectors): statistics and
  // coefficients are indexed by the padded channel c, the parameters by p = cmap[c]; -1 marks padding inside the range
  const int p = (c >= C) ? -1 : (cmap ? cmap[c] : c);
  if (p < 0) {  // padding lanes of the channel vectors stay neutral
    scale[c] = 0.f; shift[c] = 0.f;
    if (save_mean) { save_mean[c] = 0.f; save_invstd[c] = 0.f; }
    return;
  }
  const float mean = a0 * inv_count;
  float var = a1 * inv_count - mean * mean;
  var = fmaxf(var, 0.f);
  const float invstd = 1.0f / sqrtf(var + eps);
  const float g = gamma ? gamma[p] : 1.f, b = beta ? beta[p] : 0.f;
  const float s = g * invstd;
  scale[c] = s;
  shift[c] = b - mean * s;
  if (save_mean) { save_mean[c] = mean; save_invstd[c] = invstd; }
  if (running_mean) {
    float m = momentum;
    if (m < 0.f) m = 1.0f / (float)(nbt[0] + 1);  // cumulative moving average; the caller bumps the counter afterwards
    running_mean[p] = (1.f - m) * running_mean[p] + m * mean;
    running_var[p] = (1.f - m) * running_var[p] + m * var * unbias;
  }
}

__global__ void k_bn_eval_coeffs(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                 const float* __restrict__ rv, float eps, float* __restrict__ scale, float* __restrict__ shift, int C,
                                 int Cpad, const int* __restrict__ cmap) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cpad) return;
  const int p = (c >= C) ? -1 : (cmap ? cmap[c] : c);
  if (p < 0) { scale[c] = 0.f; shift[c] = 0.f; return; }
  const float invstd = 1.0f / sqrtf(rv[p] + eps);
  const float s = (gamma ? gamma[p] : 1.f) * invstd;
  scale[c] = s;
  shift[c] = (beta ? beta[p] : 0.f) - rm[p] * s;
}

// stats2 = [sum g, sum g*x];  dgamma = invstd*(sum g*x - mean*sum g), dbeta = sum g,
// dx = c1*g + c2*x + c3 with c1 = gamma*invstd, c2 = -gamma*invstd^2*dgamma/M, c3 = gamma*invstd*(mean*invstd*dgamma - dbeta)/M
__global__ __launch_bounds__(256) void k_bn_finalize_bwd(const float* __restrict__ stats2, int rows, int stat_ld, float inv_count,
                                  const float* __restrict__ gamma,
                                  const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                  const float* __restrict__ rho_ptr, const float* __restrict__ penalty, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta, float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ c3,
                                  int C, int Cpad, const int* __restrict__ cmap) {
  __shared__ float s_red[2][64][17];
  const int c = blockIdx.x * 16 + threadIdx.x;
  float sg, sgx;
  stat_row_sum(stats2, rows, stat_ld, Cpad, s_red, sg, sgx);
  if (threadIdx.x >= 16 || c >= Cpad) return;
  const int p = (c >= C) ? -1 : (cmap ? cmap[c] : c);   // parameter index (see k_bn_finalize_fwd)
  if (p < 0) { c1[c] = 0.f; c2[c] = 0.f; c3[c] = 0.f; return; }
  const float mean = save_mean[c], r = save_invstd[c];
  const float g = gamma ? gamma[p] : 1.f;
  const float dg = r * (sgx - mean * sg);
  const float db = sg;
  c1[c] = g * r;
  c2[c] = -g * r * r * dg * inv_count;
  c3[c] = g * r * (mean * r * dg - db) * inv_count;
  if (dgamma) {
    float l1 = 0.f;
    if (rho_ptr && penalty) {
      const float s = (g > 0.f) ? 1.f : ((g < 0.f) ? -1.f : 0.f);
      l1 = rho_ptr[0] * penalty[p] * s;
    }
    dgamma[p] += dg + l1;  // gradients accumulate into the (zeroed) arena, like autograd's AccumulateGrad
  }
  if (dbeta) dbeta[p] += db;
}

// y = act(x*scale + shift) (+ res), 8 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void k_bn_apply(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, int relu, const T* __restrict__ res, int ldres,
                                                  T* __restrict__ y, int ldy, long M, int C) {
  const int cg = (C + 7) / 8;
  const long total = M * cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / cg;
    const int c0 = (int)(i % cg) * 8;
    float v[8], s[8], h[8];
    VecIO<T, 8>::load(x + m * ldx + c0, v);
    VecIO<float, 8>::load(scale + c0, s);
    VecIO<float, 8>::load(shift + c0, h);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = v[e] * s[e] + h[e];
      a = act_apply(a, act_of(relu));
      v[e] = a;
    }
    if (res) {
      float r[8];
      VecIO<T, 8>::load(res + m * ldres + c0, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e >= C) v[e] = 0.f;
    VecIO<T, 8>::store(y + m * ldy + c0, v);
  }
}

// dP = c1*g + c2*x + c3 per channel, rounded to the storage type: the differentiated BatchNorm output as a tensor.  The GEMM
// prologue PRO_BNBWD computes exactly this (same expression, same rounding) per consumer tile; for the 14x14 / 7x7 stages the
// projection's input-gradient GEMM has 23..54 consumers (64-channel chunks) of every row, so it is cheaper to materialise the
// narrow tensor once (10 MB) and run the streaming GEMM (k_gemm_nt_st) without a prologue.
template <typename T>
__global__ __launch_bounds__(256) void k_bnbwd_apply(const T* __restrict__ g, int ldg, const T* __restrict__ x, int ldx,
                                                     const float* __restrict__ c1, const float* __restrict__ c2,
                                                     const float* __restrict__ c3, T* __restrict__ y, int ldy, long M, int C) {
  const int cg = (C + 7) / 8;
  const long total = M * cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / cg;
    const int c0 = (int)(i % cg) * 8;
    float v[8], xx[8], a1[8], a2[8], a3[8];
    VecIO<T, 8>::load(g + m * ldg + c0, v);
    VecIO<T, 8>::load(x + m * ldx + c0, xx);
    VecIO<float, 8>::load(c1 + c0, a1);
    VecIO<float, 8>::load(c2 + c0, a2);
    VecIO<float, 8>::load(c3 + c0, a3);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? a1[e] * v[e] + a2[e] * xx[e] + a3[e] : 0.f;
    VecIO<T, 8>::store(y + m * ldy + c0, v);
  }
}

__device__ __forceinline__ unsigned hash32(unsigned long long key) {
  // splitmix64 finaliser; counter-based so that backward can regenerate nothing: the keep mask is stored
  key += 0x9E3779B97F4A7C15ull;
  key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull;
  key = (key ^ (key >> 27)) * 0x94D049BB133111EBull;
  key = key ^ (key >> 31);
  return (unsigned)(key >> 32);
}

// pooled[n][c] = dropout( mean_hw act(x[n,hw,c]*scale+shift) ); one thread per (n, 8 channels)
template <typename T>
__global__ __launch_bounds__(256) void k_bn_act_pool(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int relu, T* __restrict__ pooled, int ldp,
                                                     unsigned char* __restrict__ keep, float drop_p, unsigned long long seed,
                                                     const long long* __restrict__ step_ptr, int N, int HW, int C) {
  const int cg = (C + 7) / 8;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * cg) return;
  const int n = i / cg, c0 = (i % cg) * 8;
  float s[8], h[8], acc[8];
  VecIO<float, 8>::load(scale + c0, s);
  VecIO<float, 8>::load(shift + c0, h);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const T* xp = x + (long)n * HW * ldx + c0;
  for (int p = 0; p < HW; ++p) {
    float v[8];
    VecIO<T, 8>::load(xp + (long)p * ldx, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = v[e] * s[e] + h[e];
      acc[e] += act_apply(a, act_of(relu));
    }
  }
  const float inv = 1.0f / (float)HW;
  const float keep_scale = (drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.f;
  const unsigned long long step = step_ptr ? (unsigned long long)step_ptr[0] : 0ull;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float a = (c0 + e < C) ? acc[e] * inv : 0.f;
    if (drop_p > 0.f) {
      const unsigned r = hash32(seed ^ (step * 0x100000001B3ull) ^ ((unsigned long long)(n * (long)C + c0 + e) << 20));
      const bool k = ((float)(r >> 8) * (1.0f / 16777216.0f)) >= drop_p;
      if (keep && c0 + e < C) keep[(long)n * C + c0 + e] = k ? 1 : 0;
      a = k ? a * keep_scale : 0.f;
    }
    o[e] = a;
  }
  VecIO<T, 8>::store(pooled + (long)n * ldp + c0, o);
}

// g[n,hw,c] = dpooled[n][c] * keep * keep_scale / HW * [x*scale+shift > 0];  stats2 += [sum g, sum g*x]
// block = 256 threads = 32 channel-groups (8 ch) x 8 pixel lanes; grid.x over (n, hw chunks), grid.y over channel groups
template <typename T>
__global__ __launch_bounds__(256) void k_pool_act_bwd(const T* __restrict__ dpooled, int ldp, const unsigned char* __restrict__ keep,
                                                      float drop_p, const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu, T* __restrict__ g, int ldg,
                                                      float* __restrict__ stats2, int stat_rows, int N, int HW, int C) {
  __shared__ float s_red[8][512];   // [pixel lane][channel of the block][2]: combined in pixel-lane order (no atomics)
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
    const float keep_scale = (drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.f;
    const float inv = 1.0f / (float)HW;
    const long total = (long)N * HW;
    for (long p = (long)blockIdx.x * 8 + pl; p < total; p += (long)gridDim.x * 8) {
      const int n = (int)(p / HW);
      float d[8], v[8], o[8];
      VecIO<T, 8>::load(dpooled + (long)n * ldp + c0, d);
      VecIO<T, 8>::load(x + p * ldx + c0, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float gg = d[e] * inv;
        if (drop_p > 0.f && keep) gg = (c0 + e < C && keep[(long)n * C + c0 + e]) ? gg * keep_scale : 0.f;
        const float a = v[e] * s[e] + h[e];
        gg = act_bwd(gg, a, act_of(relu));
        if (c0 + e >= C) gg = 0.f;
        gg = to_f32(from_f32<T>(gg));
        o[e] = gg;
        s0[e] += gg;
        s1[e] += gg * v[e];
      }
      VecIO<T, 8>::store(g + p * ldg + c0, o);
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
      float* srow = stats2 + (long)blockIdx.x * 2 * C;   // one row per workgroup column (gridDim.x <= stat_rows)
      srow[c] = a;
      srow[C + c] = b;
      stat_zero_tail(stats2, 2L * C, blockIdx.x + gridDim.x, gridDim.x, stat_rows, c);
      stat_zero_tail(stats2, 2L * C, blockIdx.x + gridDim.x, gridDim.x, stat_rows, C + c);
    }
  }
}

// generic "masked gradient + statistics" pass:  g = dy * [z*scale+shift > 0];  stats2 rows = [sum g, sum g*z]
// block = CGL channel groups (8 channels each) x 256/CGL pixel lanes, CGL = 2^lcg chosen on the host so that narrow tensors
// (the 16..320-channel block outputs this runs on) keep all 256 threads busy; the pixel lanes are combined in lane order.
template <typename T>
__global__ __launch_bounds__(256) void k_act_bwd_stats(const T* __restrict__ dy, int lddy, const T* __restrict__ z, int ldz,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                       T* __restrict__ g, int ldg, float* __restrict__ stats2, int stat_rows, long M,
                                                       int C, int lcg) {
  __shared__ float s_red[256 * 16];   // [pixel lane][channel of the block column][2]
  const int tid = threadIdx.x;
  const int CGL = 1 << lcg, PL = 256 >> lcg;
  const int cgl = tid & (CGL - 1), pl = tid >> lcg;
  const int c0 = (blockIdx.y * CGL + cgl) * 8;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  if (c0 < C) {
    float s[8], h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 1.f; h[e] = 0.f; }
    if (scale) { VecIO<float, 8>::load(scale + c0, s); VecIO<float, 8>::load(shift + c0, h); }
    for (long p = (long)blockIdx.x * PL + pl; p < M; p += (long)gridDim.x * PL) {
      float d[8], v[8];
      VecIO<T, 8>::load(dy + p * lddy + c0, d);
      VecIO<T, 8>::load(z + p * ldz + c0, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = v[e] * s[e] + h[e];
        float gg = d[e];
        gg = act_bwd(gg, a, act_of(relu));
        if (c0 + e >= C) gg = 0.f;
        d[e] = gg;
        s0[e] += gg;
        s1[e] += gg * v[e];
      }
      if (g) VecIO<T, 8>::store(g + p * ldg + c0, d);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s_red[(pl * CGL * 8 + cgl * 8 + e) * 2] = s0[e];
    s_red[(pl * CGL * 8 + cgl * 8 + e) * 2 + 1] = s1[e];
  }
  __syncthreads();
  if (stats2 && tid < CGL * 8) {
    const int c = blockIdx.y * CGL * 8 + tid;
    if (c < C) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < PL; ++q) { a += s_red[(q * CGL * 8 + tid) * 2]; b += s_red[(q * CGL * 8 + tid) * 2 + 1]; }
      float* srow = stats2 + (long)blockIdx.x * 2 * C;   // one row per workgroup column (gridDim.x <= stat_rows)
      srow[c] = a;
      srow[C + c] = b;
      stat_zero_tail(stats2, 2L * C, blockIdx.x + gridDim.x, gridDim.x, stat_rows, c);
      stat_zero_tail(stats2, 2L * C, blockIdx.x + gridDim.x, gridDim.x, stat_rows, C + c);
    }
  }
}

}  // namespace atomnas

using namespace atomnas;

extern "C" int atomnas_bn_finalize_fwd(const float* stats, int stat_rows, int stat_ld, double count, const float* gamma, const float* beta, float eps,
                                       float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                                       float* scale, float* shift, float* save_mean, float* save_invstd, int C, const int* cmap,
                                       void* stream) {
  ATOMNAS_REQUIRE(stats && stat_rows > 0 && scale && shift && C > 0 && count > 0, "bn_finalize_fwd: bad arguments");
  ATOMNAS_REQUIRE(stat_ld >= (C + 7) / 8 * 8 && stat_ld % 4 == 0 && ((size_t)stats & 15) == 0, "bn_finalize_fwd: the statistics rows must be 16-byte aligned with a pitch of at least C rounded up to 8 (stat_ld=%d)", stat_ld);
  ATOMNAS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_fwd: running stats must come together");
  ATOMNAS_REQUIRE(!(momentum < 0.f && running_mean) || num_batches_tracked, "bn_finalize_fwd: cumulative mode needs the batch counter");
  const int Cpad = (C + 7) / 8 * 8;
  const float unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bn_finalize_fwd, dim3((Cpad + 15) / 16), dim3(256), 0, st, stats, stat_rows, stat_ld, (float)(1.0 / count), unbias, gamma, beta,
                     eps, momentum, running_mean, running_var, num_batches_tracked, scale, shift, save_mean, save_invstd, C, Cpad, cmap);
  return check_launch("bn_finalize_fwd");
}

extern "C" int atomnas_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                      float eps, float* scale, float* shift, int C, const int* cmap, void* stream) {
  ATOMNAS_REQUIRE(running_mean && running_var && scale && shift && C > 0, "bn_eval_coeffs: bad arguments");
  const int Cpad = (C + 7) / 8 * 8;
  hipLaunchKernelGGL(k_bn_eval_coeffs, dim3((Cpad + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, scale, shift, C, Cpad, cmap);
  return check_launch("bn_eval_coeffs");
}

extern "C" int atomnas_bn_finalize_bwd(const float* stats2, int stat_rows, int stat_ld, double count, const float* gamma, const float* save_mean,
                                       const float* save_invstd, const float* rho_ptr, const float* penalty, float* dgamma,
                                       float* dbeta, float* c1, float* c2, float* c3, int C, const int* cmap, void* stream) {
  ATOMNAS_REQUIRE(stats2 && stat_rows > 0 && save_mean && save_invstd && c1 && c2 && c3 && C > 0 && count > 0, "bn_finalize_bwd: bad arguments");
  ATOMNAS_REQUIRE(stat_ld >= (C + 7) / 8 * 8 && stat_ld % 4 == 0 && ((size_t)stats2 & 15) == 0, "bn_finalize_bwd: the statistics rows must be 16-byte aligned with a pitch of at least C rounded up to 8 (stat_ld=%d)", stat_ld);
  const int Cpad = (C + 7) / 8 * 8;
  hipLaunchKernelGGL(k_bn_finalize_bwd, dim3((Cpad + 15) / 16), dim3(256), 0, (hipStream_t)stream, stats2, stat_rows, stat_ld, (float)(1.0 / count),
                     gamma, save_mean, save_invstd, rho_ptr, penalty, dgamma, dbeta, c1, c2, c3, C, Cpad, cmap);
  return check_launch("bn_finalize_bwd");
}

extern "C" int atomnas_bn_apply(const void* x, int ldx, const float* scale, const float* shift, int relu, const void* res, int ldres,
                                void* y, int ldy, long M, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && y && scale && shift && M > 0 && C > 0, "bn_apply: bad arguments");
  ATOMNAS_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C && (!res || (ldres % 8 == 0 && ldres >= C)), "bn_apply: bad pitch");
  const long total = M * ((C + 7) / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_bn_apply<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, ldx, scale, shift, relu,
                       (const float*)res, ldres, (float*)y, ldy, M, C);
  else
    hipLaunchKernelGGL(k_bn_apply<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, ldx, scale, shift, relu,
                       (const bf16_t*)res, ldres, (bf16_t*)y, ldy, M, C);
  return check_launch("bn_apply");
}

extern "C" int atomnas_bnbwd_apply(const void* g, int ldg, const void* x, int ldx, const float* c1, const float* c2, const float* c3,
                                   void* y, int ldy, long M, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(g && x && y && c1 && c2 && c3 && M > 0 && C > 0, "bnbwd_apply: bad arguments");
  ATOMNAS_REQUIRE(ldg % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldg >= C && ldx >= C && ldy >= C, "bnbwd_apply: bad pitch");
  // the kernel moves whole 8-channel groups: c1 / c2 / c3 must be readable up to C rounded up to 8 (include/atomnas_hip.h), the tensors
  // are accessed 16 bytes at a time
  ATOMNAS_REQUIRE((((size_t)g | (size_t)x | (size_t)y) & 15) == 0 && (((size_t)c1 | (size_t)c2 | (size_t)c3) & 15) == 0,
                  "bnbwd_apply: g, x, y and the coefficient vectors must be 16-byte aligned");
  const long total = M * ((C + 7) / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_bnbwd_apply<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)g, ldg, (const float*)x, ldx, c1, c2,
                       c3, (float*)y, ldy, M, C);
  else
    hipLaunchKernelGGL(k_bnbwd_apply<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)g, ldg, (const bf16_t*)x, ldx, c1,
                       c2, c3, (bf16_t*)y, ldy, M, C);
  return check_launch("bnbwd_apply");
}

extern "C" int atomnas_bn_act_pool(const void* x, int ldx, const float* scale, const float* shift, int relu, void* pooled, int ldp,
                                   unsigned char* keep, float drop_p, unsigned long long seed, const long long* step_ptr, int N,
                                   int HW, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && pooled && scale && shift && N > 0 && HW > 0 && C > 0, "bn_act_pool: bad arguments");
  ATOMNAS_REQUIRE(ldx % 8 == 0 && ldp % 8 == 0 && ldx >= C && ldp >= C, "bn_act_pool: bad pitch");
  ATOMNAS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "bn_act_pool: bad dropout ratio");
  const int total = N * ((C + 7) / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_bn_act_pool<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)x, ldx, scale, shift, relu,
                       (float*)pooled, ldp, keep, drop_p, seed, step_ptr, N, HW, C);
  else
    hipLaunchKernelGGL(k_bn_act_pool<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)x, ldx, scale, shift, relu,
                       (bf16_t*)pooled, ldp, keep, drop_p, seed, step_ptr, N, HW, C);
  return check_launch("bn_act_pool");
}

extern "C" int atomnas_pool_act_bwd(const void* dpooled, int ldp, const unsigned char* keep, float drop_p, const void* x, int ldx,
                                    const float* scale, const float* shift, int relu, void* g, int ldg, float* stats2, int stat_rows,
                                    int N, int HW, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(dpooled && x && g && scale && shift && N > 0 && HW > 0 && C > 0, "pool_act_bwd: bad arguments");
  ATOMNAS_REQUIRE(ldx % 8 == 0 && ldp % 8 == 0 && ldg % 8 == 0 && ldx >= C && ldp >= C && ldg >= C, "pool_act_bwd: bad pitch");
  ATOMNAS_REQUIRE(!stats2 || stat_rows > 0, "pool_act_bwd: statistics need stat_rows > 0");
  long gx = ((long)N * HW + 7) / 8;
  if (gx > 1024) gx = 1024;
  if (stats2 && gx > stat_rows) gx = stat_rows;
  dim3 grid((unsigned)gx, (C + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_pool_act_bwd<float>, grid, dim3(256), 0, st, (const float*)dpooled, ldp, keep, drop_p, (const float*)x, ldx,
                       scale, shift, relu, (float*)g, ldg, stats2, stat_rows, N, HW, C);
  else
    hipLaunchKernelGGL(k_pool_act_bwd<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dpooled, ldp, keep, drop_p, (const bf16_t*)x,
                       ldx, scale, shift, relu, (bf16_t*)g, ldg, stats2, stat_rows, N, HW, C);
  return check_launch("pool_act_bwd");
}

extern "C" int atomnas_act_bwd_stats(const void* dy, int lddy, const void* z, int ldz, const float* scale, const float* shift,
                                     int relu, void* g, int ldg, float* stats2, int stat_rows, long M, int C, int dtype, void* stream) {
  ATOMNAS_REQUIRE(dy && z && M > 0 && C > 0, "act_bwd_stats: bad arguments");
  ATOMNAS_REQUIRE(lddy % 8 == 0 && ldz % 8 == 0 && lddy >= C && ldz >= C && (!g || (ldg % 8 == 0 && ldg >= C)), "act_bwd_stats: bad pitch");
  ATOMNAS_REQUIRE((scale == nullptr) == (shift == nullptr), "act_bwd_stats: scale/shift must come together");
  ATOMNAS_REQUIRE(!stats2 || stat_rows > 0, "act_bwd_stats: statistics need stat_rows > 0");
  const int ncg = (C + 7) / 8;
  int lcg = 0;
  while ((1 << lcg) < ncg && lcg < 5) ++lcg;   // channel groups per block column: 1, 2, 4, ..., 32
  const int cgl = 1 << lcg, pl = 256 >> lcg;
  long gx = (M + pl - 1) / pl;
  if (gx > 2048) gx = 2048;
  if (stats2 && gx > stat_rows) gx = stat_rows;
  dim3 grid((unsigned)gx, (ncg + cgl - 1) / cgl);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(k_act_bwd_stats<float>, grid, dim3(256), 0, st, (const float*)dy, lddy, (const float*)z, ldz, scale, shift,
                       relu, (float*)g, ldg, stats2, stat_rows, M, C, lcg);
  else
    hipLaunchKernelGGL(k_act_bwd_stats<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, lddy, (const bf16_t*)z, ldz, scale, shift,
                       relu, (bf16_t*)g, ldg, stats2, stat_rows, M, C, lcg);
  return check_launch("act_bwd_stats");
}

// Optimizer tail of the training step on flat fp32 arenas: one launch replaces the ~12k tiny ATen dispatches of
//   cal_l2_loss 'mnas'        (utils/optim.py:226-243,249)   grad += wd * p on conv/fc weights and the classifier bias
//   RMSprop.step              (utils/rmsprop.py:70-132)      TF-style, eps inside the sqrt, momentum buffer, lr outside
//   ExponentialMovingAverage  (utils/optim.py:54-65)         shadow = d*shadow + (1-d)*p, d = min(decay, (1+n)/(10+n))
// plus the re-packing of the updated weights into the layouts the convolution kernels consume.
// Scalars that change every iteration (lr, EMA decay, 1/world) are read from device memory so that the launch can sit
// inside a replayed hipGraph.
#include "common.h"

namespace atomnas {

enum { HYP_LR = 0, HYP_RHO = 1, HYP_EMA_DECAY = 2, HYP_GRAD_SCALE = 3 };

// arithmetic order follows the reference: sq.mul_(alpha).addcmul_(1-alpha, g, g); avg = sqrt(sq+eps);
// buf.mul_(mom).addcdiv_(g, avg); p.add_(-lr, buf)
__global__ __launch_bounds__(256) void k_rmsprop_ema(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                     float* __restrict__ buf, float* __restrict__ ema,
                                                     const float* __restrict__ wd_chunk, long n, const float* __restrict__ hyper,
                                                     float alpha, float one_minus_alpha, float eps, int eps_inside_sqrt,
                                                     float momentum, float* __restrict__ l2_part) {
#pragma clang fp contract(off)  // keep the reference's separate roundings (no fused multiply-add)
  __shared__ float s_part[4];
  const float lr = hyper[HYP_LR], d = hyper[HYP_EMA_DECAY], gs = hyper[HYP_GRAD_SCALE];
  const long nchunks = (n + 255) / 256;
  float l2acc = 0.f;
  for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const long i = ch * 256 + threadIdx.x;
    if (i >= n) continue;
    const float wd = wd_chunk ? wd_chunk[ch] : 0.f;
    float pv = p[i];
    l2acc += (wd * pv) * pv;   // value of the L2 regulariser at the weights this step's loss saw (before the update)
    float gv = g[i] * gs;
    gv = gv + wd * pv;
    float s = sq[i] * alpha;
    s = s + (one_minus_alpha * gv) * gv;
    sq[i] = s;
    const float avg = eps_inside_sqrt ? sqrtf(s + eps) : (sqrtf(s) + eps);
    if (buf) {
      float b = buf[i] * momentum;
      b = b + gv / avg;
      buf[i] = b;
      pv = pv - lr * b;
    } else {
      pv = pv - lr * (gv / avg);
    }
    p[i] = pv;
    if (ema && d >= 0.f) ema[i] = ema[i] * d + (1.0f - d) * pv;
  }
  if (l2_part) {   // per-workgroup partial, fixed order inside the workgroup; summed by k_sum_partials
    l2acc = wave_sum(l2acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = l2acc;
    __syncthreads();
    if (threadIdx.x == 0) l2_part[blockIdx.x] = 0.5f * (((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]);
  }
}

// one workgroup: out[0] (+)= scale * sum of n values in a fixed order (thread t adds t, t+256, ...; then a fixed tree)
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ ws, int n, float scale, int accumulate,
                                                      float* __restrict__ out) {
  __shared__ float s_t[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += ws[i];
  s_t[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_t[threadIdx.x] += s_t[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * s_t[0];
}

__global__ __launch_bounds__(256) void k_ema(float* __restrict__ shadow, const float* __restrict__ x, long n,
                                             const float* __restrict__ hyper) {
#pragma clang fp contract(off)
  const float d = hyper[HYP_EMA_DECAY];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) shadow[i] = shadow[i] * d + (1.0f - d) * x[i];
}

// x *= hyper[idx]  (the BN running statistics summed over the ranks -> their average: hyper[HYP_GRAD_SCALE] = 1 / world)
__global__ __launch_bounds__(256) void k_scale_by(float* __restrict__ x, long n, const float* __restrict__ hyper, int idx) {
  const float f = hyper[idx];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= f;
}

// zero-fill with 16-byte stores (n16 pieces) plus a 4-byte tail.  A kernel rather than hipMemsetAsync: as a captured memset node
// the 44 MB gradient arena was not cleared on replay (bs 256 supernet, ROCm 7.2: the replayed step trained on accumulated
// gradients), while kernel nodes replay faithfully.
__global__ __launch_bounds__(256) void k_zero(f32x4* __restrict__ p16, long n16, float* __restrict__ tail, int ntail) {
  const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p16[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}

__global__ __launch_bounds__(256) void k_add_i64(long* __restrict__ p, long n, long v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] += v;
}

// Weight packing jobs.  src is fp32 in the parameter arena with logical shape [rows][cols] (row pitch src_ld).
//   mode 0 (PW):   dst[r*dst_ld + c_off + c] = src[r][c]            (storage T)   -> gemm_nt weight  [N][K]
//   mode 1 (PW_T): dst[(c_off + c)*dst_ld + r] = src[r][c]          (storage T)   -> gemm_nt weight of the transposed product
//   mode 2 (DW):   dst[c*dst_ld + c_off + r] = src[r][c]            (fp32)        -> depthwise taps [k*k][C]
struct PackJob {
  long src_off, dst_off;
  int rows, cols, src_ld, dst_ld, c_off, mode;
};

template <typename T>
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ arena, void* __restrict__ packbuf, const PackJob* __restrict__ jobs) {
  const PackJob jb = jobs[blockIdx.y];
  const long total = (long)jb.rows * jb.cols;
  const float* src = arena + jb.src_off;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / jb.cols), c = (int)(i % jb.cols);
    const float v = src[(long)r * jb.src_ld + c];
    if (jb.mode == 0) reinterpret_cast<T*>(packbuf)[jb.dst_off + (long)r * jb.dst_ld + jb.c_off + c] = from_f32<T>(v);
    else if (jb.mode == 1) reinterpret_cast<T*>(packbuf)[jb.dst_off + (long)(jb.c_off + c) * jb.dst_ld + r] = from_f32<T>(v);
    else reinterpret_cast<float*>(packbuf)[jb.dst_off + (long)c * jb.dst_ld + jb.c_off + r] = v;
  }
}

}  // namespace atomnas

using namespace atomnas;

extern "C" int atomnas_fused_rmsprop_ema(float* p, const float* g, float* sq, float* buf, float* ema, const float* wd_chunk, long n,
                                         const float* hyper, double alpha, double eps, int eps_inside_sqrt, double momentum,
                                         float* l2_value, float* ws, void* stream) {
  ATOMNAS_REQUIRE(p && g && sq && hyper && n > 0, "fused_rmsprop_ema: bad arguments");
  ATOMNAS_REQUIRE(!l2_value || (ws && wd_chunk), "fused_rmsprop_ema: the L2 value needs wd_chunk and a 4096-float workspace");
  ATOMNAS_REQUIRE(momentum >= 0.0 && alpha >= 0.0 && eps >= 0.0, "fused_rmsprop_ema: bad hyper-parameters");
  ATOMNAS_REQUIRE((momentum > 0.0) == (buf != nullptr), "fused_rmsprop_ema: momentum buffer must be given iff momentum > 0");
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_rmsprop_ema, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, sq, buf, ema, wd_chunk, n, hyper,
                     (float)alpha, (float)(1.0 - alpha), (float)eps, eps_inside_sqrt, (float)momentum, l2_value ? ws : nullptr);
  if (l2_value)
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, (int)blocks, 1.f, 0, l2_value);
  return check_launch("fused_rmsprop_ema");
}

// out[0] = scale * sum_i x[i] in a fixed order (mean of the per-sample losses, train.py:178-180 `loss = torch.mean(loss)`)
extern "C" int atomnas_vec_sum(const float* x, int n, float scale, float* out, void* stream) {
  ATOMNAS_REQUIRE(x && out && n > 0, "vec_sum: bad arguments");
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, scale, 0, out);
  return check_launch("vec_sum");
}

extern "C" int atomnas_ema_update(float* shadow, const float* x, long n, const float* hyper, void* stream) {
  ATOMNAS_REQUIRE(shadow && x && hyper && n > 0, "ema_update: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_ema, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, shadow, x, n, hyper);
  return check_launch("ema_update");
}

extern "C" int atomnas_scale_by(float* x, long n, const float* hyper, int idx, void* stream) {
  ATOMNAS_REQUIRE(x && hyper && n > 0 && idx >= 0 && idx < 4, "scale_by: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_scale_by, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, hyper, idx);
  return check_launch("scale_by");
}

// p[0 .. bytes) = 0 (gradient arena, per-step scalars, accumulated outputs); p 16-byte aligned, bytes a multiple of 4
extern "C" int atomnas_zero(void* p, long bytes, void* stream) {
  ATOMNAS_REQUIRE(p && bytes > 0 && bytes % 4 == 0 && ((unsigned long long)p & 15ull) == 0, "zero: needs a 16-byte aligned pointer and whole words");
  const long n16 = bytes / 16;
  const int ntail = (int)((bytes % 16) / 4);
  long blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_zero, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (f32x4*)p, n16, (float*)p + n16 * 4, ntail);
  return check_launch("zero");
}

// p[i] += v for the int64 counters (num_batches_tracked of every BatchNorm, models/mobilenet_base.py:142; the dropout step counter)
extern "C" int atomnas_add_i64(long* p, long n, long v, void* stream) {
  ATOMNAS_REQUIRE(p && n > 0, "add_i64: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_add_i64, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, n, v);
  return check_launch("add_i64");
}

extern "C" int atomnas_pack_weights(const float* arena, void* packbuf, const void* jobs_dev, int njobs, int dtype, void* stream) {
  ATOMNAS_REQUIRE(arena && packbuf && jobs_dev && njobs > 0, "pack_weights: bad arguments");
  dim3 grid(32, njobs);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32) hipLaunchKernelGGL(k_pack<float>, grid, dim3(256), 0, st, arena, packbuf, (const PackJob*)jobs_dev);
  else hipLaunchKernelGGL(k_pack<bf16_t>, grid, dim3(256), 0, st, arena, packbuf, (const PackJob*)jobs_dev);
  return check_launch("pack_weights");
}

// ---- regularisers as gradient contributions (so that p.grad after backward() is what the reference's autograd leaves
// there): cal_l2_loss (utils/optim.py:210-249) contributes wd*p, cal_bn_l1_loss (utils/prune.py:161-167) contributes
// rho*penalty*sign(gamma).  One launch over a job table instead of ~2.2k ATen dispatches forward and ~3.3k backward.
namespace atomnas {
struct RegJob {
  long off;
  int count;
  float coef;
};

__global__ __launch_bounds__(256) void k_reg_grad(const float* __restrict__ p, float* __restrict__ g, const RegJob* __restrict__ jobs,
                                                  int use_sign, const float* __restrict__ mult_ptr, const float* __restrict__ go_ptr) {
  const RegJob jb = jobs[blockIdx.y];
  float m = jb.coef;
  if (mult_ptr) m *= mult_ptr[0];
  if (go_ptr) m *= go_ptr[0];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.count; i += gridDim.x * 256) {
    const float v = p[jb.off + i];
    const float d = use_sign ? ((v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f)) : v;
    g[jb.off + i] += m * d;
  }
}

// stage 1: ws[job * 64 + block] = coef * post_scale * mult * (block's share of the job's sum), fixed order inside the block
__global__ __launch_bounds__(256) void k_reg_value(const float* __restrict__ p, const RegJob* __restrict__ jobs, int use_abs,
                                                   const float* __restrict__ mult_ptr, float post_scale, float* __restrict__ ws) {
  __shared__ float s_part[4];
  const RegJob jb = jobs[blockIdx.y];
  float acc = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.count; i += gridDim.x * 256) {
    const float v = p[jb.off + i];
    acc += use_abs ? fabsf(v) : v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = jb.coef * post_scale;
    if (mult_ptr) m *= mult_ptr[0];
    ws[blockIdx.y * gridDim.x + blockIdx.x] = m * (((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]);
  }
}

}  // namespace atomnas

// g[off+i] += coef * mult * go * (use_sign ? sign(p) : p) for every job {long off; int count; float coef;}
extern "C" int atomnas_reg_grad(const float* p, float* g, const void* jobs_dev, int njobs, int use_sign, const float* mult_ptr,
                                const float* grad_out_ptr, void* stream) {
  ATOMNAS_REQUIRE(p && g && jobs_dev && njobs > 0, "reg_grad: bad arguments");
  hipLaunchKernelGGL(atomnas::k_reg_grad, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream, p, g, (const atomnas::RegJob*)jobs_dev,
                     use_sign, mult_ptr, grad_out_ptr);
  return atomnas::check_launch("reg_grad");
}

// out += post_scale * mult * sum_jobs coef * sum_i (use_abs ? |p_i| : p_i^2);  ws: caller-owned scratch of 64 * njobs floats
// (per-workgroup partials, summed in a fixed order: the logged regulariser values are bit-reproducible)
extern "C" int atomnas_reg_value(const float* p, const void* jobs_dev, int njobs, int use_abs, const float* mult_ptr,
                                 float post_scale, float* out, float* ws, void* stream) {
  ATOMNAS_REQUIRE(p && out && ws && jobs_dev && njobs > 0, "reg_value: bad arguments");
  hipLaunchKernelGGL(atomnas::k_reg_value, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream, p, (const atomnas::RegJob*)jobs_dev,
                     use_abs, mult_ptr, post_scale, ws);
  hipLaunchKernelGGL(atomnas::k_sum_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, 64 * njobs, 1.f, 1, out);
  return atomnas::check_launch("reg_value");
}

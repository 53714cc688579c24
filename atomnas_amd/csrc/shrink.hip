// Dynamic network shrinkage on device: alive masks of all prunable BatchNorms in one launch (mask bytes, ascending
// kept-channel indices, kept counts) and the index-packed gather that rebuilds weights, optimizer state and EMA shadows.
//   masks : train.py:46-63 and utils/prune.py:190-195   mask = |gamma| > thr  (optionally OR / replaced by the EMA gamma)
//   repack: models/compress_utils.py:31-37 (_mask_along_dim) applied by compress_conv/compress_bn (:73-119) and mirrored
//           into RMSprop state (utils/rmsprop.py:134-165) and EMA shadows (utils/optim.py:134-153)
// Masks, counts and gather indices are integer results and are bit-exact with the reference by construction
// (plain fp32 compare against the threshold rounded to fp32, as torch does for `tensor > python_float`).
#include "common.h"

namespace atomnas {

struct MaskJob {
  long off;      // offset of this gamma vector inside the parameter arena (and the EMA arena)
  int count;     // channels
  int out_off;   // offset into the mask / index outputs
};

// one block per job: mask, kept count, and the gather index (ascending channel order) via a block-wide scan
__global__ __launch_bounds__(256) void k_gamma_mask(const float* __restrict__ p, const float* __restrict__ ema,
                                                    const MaskJob* __restrict__ jobs, float thr, int mode /*0 cur, 1 cur|ema, 2 ema*/,
                                                    unsigned char* __restrict__ mask, int* __restrict__ index, int* __restrict__ kept) {
  __shared__ int s_scan[256];
  __shared__ int s_base;
  const MaskJob jb = jobs[blockIdx.x];
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < jb.count; c0 += 256) {
    const int c = c0 + threadIdx.x;
    int alive = 0;
    if (c < jb.count) {
      const bool a = fabsf(p[jb.off + c]) > thr;
      const bool b = ema ? (fabsf(ema[jb.off + c]) > thr) : false;
      alive = (mode == 0) ? a : ((mode == 1) ? (a || b) : b);
      mask[jb.out_off + c] = (unsigned char)alive;
    }
    s_scan[threadIdx.x] = alive;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
      int v = (threadIdx.x >= o) ? s_scan[threadIdx.x - o] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += v;
      __syncthreads();
    }
    const int incl = s_scan[threadIdx.x];
    const int base = s_base;
    if (alive) index[jb.out_off + base + incl - 1] = c;
    __syncthreads();
    if (threadIdx.x == 255) s_base = base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) kept[blockIdx.x] = s_base;
}

}  // namespace atomnas

using namespace atomnas;

extern "C" int atomnas_gamma_mask(const float* params, const float* ema, const void* jobs_dev, int njobs, float threshold, int mode,
                                  unsigned char* mask, int* index, int* kept, void* stream) {
  ATOMNAS_REQUIRE(params && jobs_dev && njobs > 0 && mask && index && kept, "gamma_mask: bad arguments");
  ATOMNAS_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || ema), "gamma_mask: mode %d needs the EMA arena", mode);
  hipLaunchKernelGGL(k_gamma_mask, dim3(njobs), dim3(256), 0, (hipStream_t)stream, params, ema, (const MaskJob*)jobs_dev, threshold, mode,
                     mask, index, kept);
  return check_launch("gamma_mask");
}

// ---- single-tensor forms used by the reference-compatible per-tensor protocol (info['mask_hook'](new, old, mask))
namespace atomnas {
// index[k] = position of the k-th non-zero byte of mask[0..count); kept[0] = number of non-zero bytes (one workgroup)
__global__ __launch_bounds__(256) void k_mask_index(const unsigned char* __restrict__ mask, int count, int* __restrict__ index,
                                                    int* __restrict__ kept) {
  __shared__ int s_scan[256];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < count; c0 += 256) {
    const int c = c0 + threadIdx.x;
    const int alive = (c < count && mask[c] != 0) ? 1 : 0;
    s_scan[threadIdx.x] = alive;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int v = (threadIdx.x >= o) ? s_scan[threadIdx.x - o] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += v;
      __syncthreads();
    }
    const int incl = s_scan[threadIdx.x];
    const int base = s_base;
    if (alive) index[base + incl - 1] = c;
    __syncthreads();
    if (threadIdx.x == 255) s_base = base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) kept[0] = s_base;
}

// dst[o*dst_os + j*dst_ds + i] = src[o*src_os + index[j]*src_ds + i]   (fp32 elements)
__global__ __launch_bounds__(256) void k_gather_dim(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ index,
                                                    long src_os, long src_ds, long dst_os, long dst_ds, int outer, int n_kept, int inner) {
  const long total = (long)outer * n_kept * inner;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int i = (int)(e % inner);
    const int j = (int)((e / inner) % n_kept);
    const int o = (int)(e / ((long)inner * n_kept));
    dst[o * dst_os + j * dst_ds + i] = src[o * src_os + (long)index[j] * src_ds + i];
  }
}
}  // namespace atomnas

// ---- job-list repack (SURVEY K12 / 8(b) `channel_repack (job list)`; models/compress_utils.py:31-37, utils/rmsprop.py:134-165,
// utils/optim.py:134-153): every gather of one shrink -- model weights, BatchNorm vectors, RMSprop state, EMA shadows -- as ONE launch
// over a table in device memory.  A workgroup finds its job by binary search over the jobs' first block numbers; a job is
// k_gather_dim's (index == NULL: the identity, a plain strided copy -- the shared pw_bn keeps every channel).
namespace atomnas {
struct GatherJob {
  const float* src;
  float* dst;
  const int* index;
  long src_os, src_ds, dst_os, dst_ds;
  int outer, n_kept, inner;
  unsigned blk0;   // first workgroup of the job (ascending over the table)
};
static_assert(sizeof(GatherJob) == 72, "atomnas_gather_job layout");

__global__ __launch_bounds__(256) void k_gather_jobs(const GatherJob* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {   // the last job whose first block is <= blockIdx.x (uniform)
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const GatherJob jb = jobs[lo];
  const long total = (long)jb.outer * jb.n_kept * jb.inner;
  const long e = (long)(blockIdx.x - jb.blk0) * 256 + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e % jb.inner);
  const int j = (int)((e / jb.inner) % jb.n_kept);
  const int o = (int)(e / ((long)jb.inner * jb.n_kept));
  const long sj = jb.index ? (long)jb.index[j] : (long)j;
  jb.dst[o * jb.dst_os + j * jb.dst_ds + i] = jb.src[o * jb.src_os + sj * jb.src_ds + i];
}
}  // namespace atomnas

// jobs_dev: device array of njobs atomnas_gather_job (include/atomnas_hip.h) with blk0 filled in ascending order, one workgroup per 256
// elements of a job; nblocks = the sum.  Jobs must not overlap in their destinations (a shrink writes every new tensor once).
extern "C" int atomnas_gather_jobs(const void* jobs_dev, int njobs, long nblocks, void* stream) {
  ATOMNAS_REQUIRE(jobs_dev && njobs > 0 && nblocks > 0 && nblocks < (1L << 31), "gather_jobs: bad arguments");
  hipLaunchKernelGGL(atomnas::k_gather_jobs, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const atomnas::GatherJob*)jobs_dev, njobs);
  return atomnas::check_launch("gather_jobs");
}

extern "C" int atomnas_mask_index(const unsigned char* mask, int count, int* index, int* kept, void* stream) {
  ATOMNAS_REQUIRE(mask && index && kept && count > 0, "mask_index: bad arguments");
  hipLaunchKernelGGL(atomnas::k_mask_index, dim3(1), dim3(256), 0, (hipStream_t)stream, mask, count, index, kept);
  return atomnas::check_launch("mask_index");
}

extern "C" int atomnas_gather_dim(const float* src, float* dst, const int* index, long src_os, long src_ds, long dst_os, long dst_ds,
                                  int outer, int n_kept, int inner, void* stream) {
  ATOMNAS_REQUIRE(src && dst && index && outer > 0 && n_kept > 0 && inner > 0, "gather_dim: bad arguments");
  const long total = (long)outer * n_kept * inner;
  long blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(atomnas::k_gather_dim, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, index, src_os, src_ds,
                     dst_os, dst_ds, outer, n_kept, inner);
  return atomnas::check_launch("gather_dim");
}

// Fixed-order reductions shared by the kernels that used to accumulate with atomics: per-workgroup partial weight
// gradients (depthwise taps, 1x1 weight-gradient GEMM) are written to a caller-owned workspace and summed here in an
// order that depends only on the launch geometry, so that two runs of the same training step are bit-identical
// (reference semantics: autograd's AccumulateGrad adds ONE complete gradient tensor, utils/rmsprop.py:70-132 then reads it).
#include "common.h"
#include <mutex>
#include <vector>

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

// ---- deferred form.  Inside a hipGraph every kernel node costs ~5 us of dispatch however small it is; a training step of the
// supernet issues ~110 of these reductions (190 for AtomNAS-C+), each a few microseconds of work.  While deferral is on
// (atomnas_reduce_defer), reduce_parts only records its job; atomnas_reduce_flush then sums all recorded jobs with one launch per
// RJ_MAX jobs.  The job table travels BY VALUE in the kernel arguments (3.6 KB), so the launch is capturable and nothing has to be
// uploaded.  The caller keeps the partial buffers alive until the flush and flushes before anything reads the gradients.
// Every element is still summed by exactly one thread in the same order as in k_reduce_parts: results are bit-identical.
struct ReduceJob {
  const float* part;
  float* out;
  long part_stride, n, s_outer, s_inner;
  int parts, inner;
  unsigned blk0, pad_;
};
constexpr int RJ_MAX = 56;
struct ReduceBatch {
  ReduceJob j[RJ_MAX];
  int njobs;
};

__global__ __launch_bounds__(256) void k_reduce_batch(const ReduceBatch b) {
  __shared__ float s_acc[8][32];
  int ji = 0;
  while (ji + 1 < b.njobs && blockIdx.x >= b.j[ji + 1].blk0) ++ji;   // uniform: scalar loads from the kernel arguments
  const ReduceJob& jb = b.j[ji];
  const float* __restrict__ part = jb.part;
  const long part_stride = jb.part_stride, n = jb.n;
  const int parts = jb.parts;
  const int el = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const long e = (long)(blockIdx.x - jb.blk0) * 32 + el;
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
    const long o = (e / jb.inner) * jb.s_outer + (e % jb.inner) * jb.s_inner;
    jb.out[o] += t;
  }
}

static std::mutex g_rj_mu;          // backward runs on autograd's device thread, the flush on the caller's
static bool g_rj_defer = false;
static hipStream_t g_rj_stream = nullptr;   // the stream deferral was switched on for: only ITS reductions are recorded (round 5)
static std::vector<ReduceJob> g_rj;

// may two jobs touch the same output element?  (address ranges overlap, unless both write disjoint column bands of one row-major
// matrix: the per-branch segments of a fused block's projection weight gradient)
static bool rj_conflict(const ReduceJob& a, const ReduceJob& b) {
  auto hi = [](const ReduceJob& j) { return j.out + ((j.n - 1) / j.inner) * j.s_outer + (long)(j.inner - 1) * j.s_inner; };
  if (hi(a) < b.out || hi(b) < a.out) return false;
  if (a.s_inner == 1 && b.s_inner == 1 && a.s_outer == b.s_outer && a.inner <= a.s_outer && b.inner <= b.s_outer) {
    const long d = (b.out - a.out) % a.s_outer;   // column of b's first element relative to a's
    const long cb = d < 0 ? d + a.s_outer : d;
    if (cb >= a.inner && cb + b.inner <= a.s_outer) return false;
  }
  return true;
}

static int flush_jobs_locked(hipStream_t st) {
  size_t i = 0;
  while (i < g_rj.size()) {
    ReduceBatch b;
    b.njobs = 0;
    unsigned blk = 0;
    while (i < g_rj.size() && b.njobs < RJ_MAX) {
      bool clash = false;
      for (int q = 0; q < b.njobs && !clash; ++q) clash = rj_conflict(b.j[q], g_rj[i]);
      if (clash) break;
      ReduceJob j = g_rj[i++];
      j.blk0 = blk;
      blk += (unsigned)((j.n + 31) / 32);
      b.j[b.njobs++] = j;
    }
    hipLaunchKernelGGL(k_reduce_batch, dim3(blk), dim3(256), 0, st, b);
  }
  g_rj.clear();
  return check_launch("reduce_flush");
}

int reduce_parts(const float* part, long part_stride, int parts, long n, float* out, int inner, long s_outer, long s_inner,
                 hipStream_t st) {
  if (n <= 0 || parts <= 0) return 0;
  {
    std::lock_guard<std::mutex> lock(g_rj_mu);
    // a reduction launched on another stream (a second model, an evaluation job in the same process) is not this backward's: its
    // order against the producer of its partials would be lost and its workspace may be gone by the flush -- it runs at once
    if (g_rj_defer && st == g_rj_stream) {
      g_rj.push_back(ReduceJob{part, out, part_stride, n, s_outer, s_inner, parts, inner, 0u, 0u});
      return 0;
    }
  }
  const long blocks = (n + 31) / 32;
  hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)blocks), dim3(256), 0, st, part, part_stride, parts, n, out, inner, s_outer,
                     s_inner);
  return check_launch("reduce_parts");
}

}  // namespace atomnas

using namespace atomnas;

extern "C" int atomnas_reduce_defer(int on, void* stream) {
  std::lock_guard<std::mutex> lock(g_rj_mu);
  int rc = 0;
  if (!on && !g_rj.empty()) rc = flush_jobs_locked((hipStream_t)stream);   // switching off never drops recorded work
  g_rj_defer = on != 0;
  g_rj_stream = (hipStream_t)stream;
  return rc;
}

extern "C" int atomnas_reduce_flush(void* stream) {
  std::lock_guard<std::mutex> lock(g_rj_mu);
  if (g_rj.empty()) return 0;
  return flush_jobs_locked((hipStream_t)stream);
}


// ---- fold jobs (ABI 9): dst[r][c] += src[r][c]; src[r][c] = 0 over a table of 2-D blocks, one launch for the whole table (or a slice).
// The fused block of AtomNAS+ keeps its expand / projection weights as ONE contiguous tensor over all kernel-size groups, as the
// reference's state_dict has them (models/mobilenet_base.py:236-254), while the kernels run on branch segments padded to whole 16-channel
// slabs: the weight gradient of a layer is computed by ONE weight-gradient GEMM into a padded scratch matrix and folded into the
// contiguous gradient arena here, segment by segment (round 5 launched one GEMM per segment: 128 launches per AtomNAS-C+ step).  The
// source is cleared on the way, so the scratch is ready for the next accumulation without a fill.
namespace atomnas {
struct FoldJob {
  float* src;
  float* dst;
  long src_ld, dst_ld;
  int rows, cols;
  unsigned blk0;   // first workgroup of the job, ascending over the table
  int pad_;
};
static_assert(sizeof(FoldJob) == 48, "atomnas_fold_job layout");

__global__ __launch_bounds__(256) void k_fold_jobs(const FoldJob* __restrict__ jobs, int njobs, unsigned blk_base) {
  const unsigned b = blockIdx.x + blk_base;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {   // the last job whose first block is <= b (uniform)
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= b) lo = mid; else hi = mid - 1;
  }
  const FoldJob jb = jobs[lo];
  const long e = (long)(b - jb.blk0) * 256 + threadIdx.x;
  if (e >= (long)jb.rows * jb.cols) return;
  const int r = (int)(e / jb.cols), c = (int)(e % jb.cols);
  float* s = jb.src + r * jb.src_ld + c;
  jb.dst[r * jb.dst_ld + c] += *s;
  *s = 0.f;
}
}  // namespace atomnas

// jobs_dev: device array of atomnas_fold_job (include/atomnas_hip.h), blk0 ascending, one workgroup per 256 elements of a job; the launch
// covers jobs first .. first + njobs - 1: blk_base = blk0 of job `first`, nblocks = the workgroups of that slice.  Blocks must not overlap.
extern "C" int atomnas_fold_jobs(const void* jobs_dev, int first, int njobs, long blk_base, long nblocks, void* stream) {
  ATOMNAS_REQUIRE(jobs_dev && first >= 0 && njobs > 0 && blk_base >= 0 && nblocks > 0 && blk_base + nblocks < (1L << 31), "fold_jobs: bad arguments");
  hipLaunchKernelGGL(atomnas::k_fold_jobs, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                     (const atomnas::FoldJob*)jobs_dev + first, njobs, (unsigned)blk_base);
  return atomnas::check_launch("fold_jobs");
}

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
// Structure (both directions): a 256-thread workgroup owns a slab of CB channels and walks over spatial tiles
// (persistent loop, so the per-channel reductions are flushed once per workgroup).  Per tile the haloed operand tile is
// staged ONCE through LDS as fp32 *after* its prologue (BN apply + ReLU, or the BN-backward affine of two streams), with
// 16-byte global loads; the k*k taps then slide over LDS rows held in registers (ds_read_b64 of a channel pair, packed
// fp32 FMAs), so each element costs one HBM read, one transform and k*k FMAs instead of k reloads and k transforms.
// LDS rows are padded so that the two tile rows a 32-lane LDS group touches fall into different bank halves.
#include "common.h"
#include <unordered_map>
#include <cstdlib>
#ifndef DW_EXP
#define DW_EXP 0  // experiment switch for tools/dwbench.py: fwd: 1 no stores, 2 no compute, 3 no LDS commit; bwd: 4 no global flush, 5 no compute, 6 no h stores, 7 no x loads, 8 no dY loads, 9 no yraw loads, 10 no FMA loop, 11 no global memory traffic
#endif

#ifndef DW_DMA
#define DW_DMA 0   // experiment (backward, bf16): the dY / yraw prefetch of the next tile goes global -> LDS directly
                   // (global_load_lds_dwordx4) instead of through 2*PF*4 staging registers.  Correct (tests pass) and it removes
                   // the spills of k = 7, but the extra raw LDS buffers cost residency: +2..4 % on 16-channel slabs, +50..90 % on
                   // 32-channel ones, -5 % only for k = 7 stride 2 (tools/dwbench.py, same box) -> off
#endif
#ifndef DW_FWD_STAGE
#define DW_FWD_STAGE 0   // experiment (forward): 1 = the output tile leaves through LDS in 16-byte pieces, 0 = every work item stores its
                         // 4-byte channel pairs directly.  Measured in situ (bs 256 step, same kernels otherwise, profiles/r02_*): staging
                         // is 3-5 % SLOWER on most layers and up to +77 % where the extra LDS costs a resident workgroup (only the
                         // 112x112 k = 7 stride-2 layer gains, -18 %) -> off
#endif
#ifndef DW_SMALL_CB7
#define DW_SMALL_CB7 32   // slab width of the k = 7 backward on 7x7 maps (64 needs more than 256 registers: 98 accumulators + prefetch)
#endif
#ifndef DW_BWD_MINB5
#define DW_BWD_MINB5 2   // experiment: minimum resident workgroups per CU the k <= 5 backward instances are compiled for
#endif
#ifndef DW_RING
#define DW_RING 1  // 1 = tiles are walked column-major and the LDS operand tile is a ring over rows: a tile below the previous one
                   // loads only its new rows (the (K-1)/S halo rows stay); 0 = every tile loads its whole haloed window
#endif
#ifndef DW_TIMING
#define DW_TIMING 0  // s_memtime phase accounting (tools/dwbench.py, experiment builds only)
#endif

namespace atomnas {

#if DW_TIMING
__device__ unsigned long long g_dw_timing[8];
#define TMARK(i)                                                       \
  {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long tn_ = __builtin_readcyclecounter();       \
    tacc[i] += tn_ - tlast;                                            \
    tlast = tn_;                                                       \
    __builtin_amdgcn_sched_barrier(0);                                 \
  }
#else
#define TMARK(i)
#endif

constexpr __host__ __device__ int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
constexpr __host__ __device__ int cdiv(int a, int b) { return -fdiv(-a, b); }
constexpr __host__ __device__ int pmod(int a, int b) { return ((a % b) + b) % b; }

// 8 consecutive channels as loaded from HBM, kept raw in registers while a tile is in flight (software prefetch)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
  }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ float get(int e) const { return (float)v[e]; }
};
template <> struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void zero() { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b; }
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
  __device__ __forceinline__ float get(int e) const { return e < 4 ? a[e] : b[e - 4]; }
};

__device__ __forceinline__ void dw_touch(const Raw8<bf16_t>& r) { asm volatile("" ::"v"(r.v)); }   // "the register is read here"
__device__ __forceinline__ void dw_touch(const Raw8<float>& r) { asm volatile("" ::"v"(r.a), "v"(r.b)); }

// one channel pair of one pixel, raw
template <typename T> struct Raw2;
template <> struct Raw2<bf16_t> {
  unsigned v;
  __device__ __forceinline__ void zero() { v = 0u; }
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const unsigned*>(p); }
  __device__ __forceinline__ float get(int e) const { return __uint_as_float(e ? (v & 0xffff0000u) : (v << 16)); }
};
template <> struct Raw2<float> {
  f32x2 v;
  __device__ __forceinline__ void zero() { v = f32x2{0.f, 0.f}; }
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const f32x2*>(p); }
  __device__ __forceinline__ float get(int e) const { return v[e]; }
};

// LDS-DMA: every active lane copies 16 bytes from its own global address to LDS at lds_wave_base + lane * 16 (wave-uniform
// base in M0; inactive lanes write nothing).  Issued through inline asm on purpose: hipcc then does not count it in its
// s_waitcnt bookkeeping, so the copy stays in flight across barriers and LDS reads until lds_dma_wait() (with the builtin
// the compiler drains it at the next LDS access; cdna_hip_programming.md "Pipelining across barriers").
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_wave_base)
               : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}

// Activation layouts (include/atomnas_hip.h): plain [M][ld] (ss == 0) or slab-major [C/16][M][16] with slab stride ss elements.
// chan_base: offset of the 8-channel group starting at channel c (a multiple of 8); pix_stride: distance between pixels.
__device__ __forceinline__ long chan_base(int c, long ss) { return ss ? (long)(c >> 4) * ss + (c & 15) : (long)c; }
__device__ __forceinline__ long pix_stride(int ld, long ss) { return ss ? 16 : (long)ld; }

struct DwGeom {
  int N, H, W, C, Ho, Wo;
  int CB;                 // channels per slab (8, 16, 32 or 64)
  int TH, TW;             // tile: output pixels (forward) / input pixels (backward)
  int tiles_y, tiles_x;   // tiles per image
  int LH, LW, RP;         // LDS operand tile: rows, cols, row pitch (floats)
  int nworkers;           // workgroups per slab (persistent loop over tiles)
  int nslabs;             // channel slabs
  int xcd;                // 1: XCD-aware workgroup decode (plain layout), 0: plain order (slab-major layout)
};

static inline int lds_pitch(int lw, int cb) {
  int rp = lw * cb;
  // rows r and r+1 are read by the two halves of a 32-lane LDS group: keep them 32 banks (dwords) apart (mod 64)
  int pad = (32 - rp % 64 + 64) % 64;
  return rp + pad;
}

// ------------------------------------------------------------------------------------------------ forward
// AM: activation specialisation of the prologue -- 0: none / ReLU by the runtime flag (the tuned code), 2: ReLU6, 3: Swish
template <typename T, int K, int S, int SW, int CB, int TM, int AM>
__global__ __launch_bounds__(256) void k_dwconv_fwd(const T* __restrict__ x, int ldx, long xss, const float* __restrict__ in_scale,
                                                    const float* __restrict__ in_shift, int in_relu,
                                                    const float* __restrict__ w, int ldw, T* __restrict__ y, int ldy, long yss,
                                                    float* __restrict__ stats, int stat_ld, int stat_rows, DwGeom g) {
  constexpr int P = (K - 1) / 2;
  constexpr int IWS = (SW - 1) * S + K;  // input columns one strip needs
  constexpr int LMAX = (TM - 1) * S + K;                           // largest LDS tile extent for this instantiation
  constexpr int PF = (LMAX * LMAX * (CB / 8) + 255) / 256;         // 16-byte loads per thread per tile
  static_assert(PF <= 32, "prefetch mask is 32 bits");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                         // [LH][RP]
  float* s_w = s_in + g.LH * g.RP;            // [K*K][CB]
  float* s_st = s_w + K * K * CB;             // [4 waves][2][CB]: per-wave partial statistics, combined in wave order
  T* s_y = reinterpret_cast<T*>(s_st + 8 * CB);   // [TH*TW][CB] the tile's output, stored with the next tile's commit

  const int tid = threadIdx.x;
  constexpr int C2 = CB / 2, CG = CB / 8;
  // Workgroup b runs on XCD b % 8 (private L2 each).  PLAIN layout (g.xcd = 1): the slabs that share 128-byte lines of a pixel
  // (64 channels) must hit the same L2 at about the same time, or every line is fetched from HBM once per slab: consecutive
  // workgroups of one XCD take the slabs of one worker (= one tile sequence).  SLAB-MAJOR layout (g.xcd = 0): slabs are disjoint
  // memory, nothing to share -- plain order, which spreads any number of workgroups evenly over the XCDs.
  const int b_xcd = blockIdx.x & 7, b_local = blockIdx.x >> 3;
  const int slab = g.xcd ? b_local % g.nslabs : (int)(blockIdx.x % g.nslabs);
  const int worker = g.xcd ? (b_local / g.nslabs) * 8 + b_xcd : (int)(blockIdx.x / g.nslabs);
  if (worker >= g.nworkers) return;
  const int c_base = slab * CB;
  const int cpad = (g.C + 7) & ~7;

  for (int i = tid; i < K * K * CB; i += 256) {
    const int t = i / CB, c = i % CB;
    s_w[i] = (c_base + c < g.C) ? w[(long)t * ldw + c_base + c] : 0.f;
  }

  // staging role: fixed channel group per thread (256 % CG == 0)
  const int cg = tid % CG;
  const bool cg_ok = c_base + cg * 8 < cpad;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
  if (in_scale && cg_ok) { VecIO<float, 8>::load(in_scale + c_base + cg * 8, sc); VecIO<float, 8>::load(in_shift + c_base + cg * 8, sh); }

  // compute role: fixed channel pair per thread (256 % C2 == 0)
  const int c2 = tid % C2;
  const int ch = c_base + 2 * c2;
  const int nstrips = g.TW / SW;
  const int nitems = C2 * g.TH * nstrips;
  float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};

  const int ntiles = g.N * g.tiles_y * g.tiles_x;
  const int npix = g.LH * g.LW;
  // tile-independent index math, done once (integer division is a ~30-instruction sequence on CDNA)
  int p_iy[PF], p_ix[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int pidx = tid / CG + i * (256 / CG);
    p_iy[i] = pidx < npix ? pidx / g.LW : -100000;   // out-of-tile slots fail every bounds test
    p_ix[i] = pidx % g.LW;
  }
  // software pipeline: the next tile's HBM loads are issued before the current tile is computed and land in registers
  Raw8<T> pf[PF];
  unsigned pfmask = 0;
  // rowmin: first tile row that has to be (re)loaded; rows below it are still in the LDS ring from the tile above
  auto issue = [&](int n, int ty, int tx, int rowmin) {
    const int hi0 = ty * g.TH * S - P, wi0 = tx * g.TW * S - P;
    const long xps = pix_stride(ldx, xss);
    const T* xn = x + (long)n * g.H * g.W * xps + chan_base(cg_ok ? c_base + cg * 8 : 0, xss);
    pfmask = 0;
#pragma unroll
    for (int i = 0; i < PF; ++i) {   // branch-free: see k_dwconv_bwd
      const int hi = hi0 + p_iy[i], wi = wi0 + p_ix[i];
      const bool ok = cg_ok && p_iy[i] >= rowmin && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
      pf[i].load(xn + (ok ? ((long)hi * g.W + wi) * xps : 0));
      pfmask |= ok ? 1u << i : 0u;
    }
  };
  auto commit = [&](int rowmin, int base) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (p_iy[i] >= rowmin) {
        float v[8];
        const bool ok = (pfmask >> i) & 1u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = pf[i].get(e) * sc[e] + sh[e];
          if constexpr (AM == ACT_RELU6) a = fminf(fmaxf(a, 0.f), 6.f);   // separate instances: the ReLU code stays as tuned
          else if constexpr (AM == ACT_SWISH) a = swish_f(a);
          else a = in_relu ? fmaxf(a, 0.f) : a;
          v[e] = ok ? a : 0.f;
        }
        int slot = p_iy[i] + base;
        if (slot >= g.LH) slot -= g.LH;
        float* d = s_in + slot * g.RP + p_ix[i] * CB + cg * 8;
        *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
    }
  };
  // work items of this thread (at most NIT): output row r, strip j
  constexpr int NIT = (C2 * TM * (TM / SW) + 255) / 256;
  int it_r[NIT], it_j[NIT];
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int it = tid + q * 256;
    const int rs = it / C2;
    it_r[q] = it < nitems ? rs / nstrips : -1;
    it_j[q] = rs % nstrips;
  }

  // output pieces of 16 bytes (pixel, 8 channels) for the staged store
  constexpr int YP = (TM * TM * CG + 255) / 256;
  int yp_r[YP], yp_c[YP];
#pragma unroll
  for (int p = 0; p < YP; ++p) {
    const int pix = (tid + 256 * p) / CG;
    yp_r[p] = pix < g.TH * g.TW ? pix / g.TW : -100000;
    yp_c[p] = pix % g.TW;
  }
  const long yps = pix_stride(ldy, yss);
  auto store_y = [&](int an, int aty, int atx) {
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      const int ho = aty * g.TH + yp_r[p], wo = atx * g.TW + yp_c[p];
      if (DW_EXP != 1 && cg_ok && yp_r[p] >= 0 && ho < g.Ho && wo < g.Wo) {
        Raw8<T> v;
        v.load(s_y + (yp_r[p] * g.TW + yp_c[p]) * CB + cg * 8);
        v.store(y + (((long)an * g.Ho + ho) * g.Wo + wo) * yps + chan_base(c_base + cg * 8, yss));
      }
    }
  };
  int sn = -1, sty = 0, stx = 0;   // tile whose output is waiting in s_y

  // tile walk: every worker owns a CONTIGUOUS range of tiles, column-major inside an image (ty fastest); (n, ty, tx) are
  // carried, not re-derived.  The LDS tile is a ring over rows: when the next tile is the one BELOW the current one, its
  // first LH - TH*S rows are the current tile's last rows and stay where they are (slot of tile row r = (r + base) mod LH);
  // only the new rows are loaded.  That removes the vertical halo from the request stream: the kernels are bound by the
  // number of L1/L2-side requests, not by HBM-side bytes.
  const int t_beg = (int)((long)worker * ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * ntiles / g.nworkers);
  int tile = t_beg;
  int ty = tile % g.tiles_y, tx = (tile / g.tiles_y) % g.tiles_x, n = tile / (g.tiles_x * g.tiles_y);
  auto advance = [&](int& an, int& aty, int& atx) {
    if (++aty == g.tiles_y) { aty = 0; if (++atx == g.tiles_x) { atx = 0; ++an; } }
  };
  const int vshift = g.TH * S;                                       // rows the window moves down per tile
  const int keep = (DW_RING && vshift < g.LH) ? g.LH - vshift : 0;   // rows shared with the tile above
  int ntx = tx, nty = ty, nn = n;
  int rowmin = 0, base = 0, nrowmin = 0, nbase = 0;
  if (tile < t_end) issue(n, ty, tx, 0);
  for (; tile < t_end; ++tile) {
    const int ho0 = ty * g.TH, wo0 = tx * g.TW;
    __syncthreads();  // previous tile fully consumed, its output complete in s_y (also orders the s_w initialisation)
#pragma unroll
    for (int i = 0; i < PF; ++i) dw_touch(pf[i]);   // every prefetched register is read here on every path
    if (DW_FWD_STAGE && sn >= 0) store_y(sn, sty, stx);
    if (DW_EXP != 3) commit(rowmin, base);
    __syncthreads();
    ntx = tx; nty = ty; nn = n;
    advance(nn, nty, ntx);
    {
      const bool below = keep > 0 && nn == n && ntx == tx && nty == ty + 1;
      nrowmin = below ? keep : 0;
      nbase = below ? base + vshift : 0;
      if (nbase >= g.LH) nbase -= g.LH;
    }
    if (tile + 1 < t_end) issue(nn, nty, ntx, nrowmin);

#pragma unroll
    for (int q = 0; q < (DW_EXP == 2 ? 0 : NIT); ++q) {
      const int r = it_r[q], j = it_j[q];
      const int ho = ho0 + r;
      if (r < 0 || ho >= g.Ho) continue;
      f32x2 acc[SW];
#pragma unroll
      for (int t = 0; t < SW; ++t) acc[t] = f32x2{0.f, 0.f};
#pragma unroll 1
      for (int ky = 0; ky < K; ++ky) {  // not unrolled: keeps one LDS row (IWS pairs) live instead of K of them
        int slot = r * S + ky + base;
        if (slot >= g.LH) slot -= g.LH;
        const float* row = s_in + slot * g.RP + (j * SW * S) * CB + 2 * c2;
        f32x2 in[IWS];
        {   // single ds_read_b64 each, see k_dwconv_bwd
          const unsigned ra = lds_addr(row);
#pragma unroll
          for (int i = 0; i < IWS; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(in[i]) : "v"(ra), "i"(i * CB * 4));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const f32x2 wv = *reinterpret_cast<const f32x2*>(s_w + (ky * K + kx) * CB + 2 * c2);
#pragma unroll
          for (int t = 0; t < SW; ++t) acc[t] += in[t * S + kx] * wv;
        }
      }
      if (ch < cpad) {
        T* yr = y + (((long)n * g.Ho + ho) * g.Wo) * yps + chan_base(ch & ~7, yss) + (ch & 7);
        T* sy = s_y + (r * g.TW + j * SW) * CB + 2 * c2;
#pragma unroll
        for (int t = 0; t < SW; ++t) {
          const int wo = wo0 + j * SW + t;
          if (wo < g.Wo) {
            float o[2];
            o[0] = (ch < g.C) ? to_f32(from_f32<T>(acc[t][0])) : 0.f;
            o[1] = (ch + 1 < g.C) ? to_f32(from_f32<T>(acc[t][1])) : 0.f;
            if (DW_FWD_STAGE) VecIO<T, 2>::store(sy + t * CB, o);
            else if (DW_EXP != 1) VecIO<T, 2>::store(yr + (long)wo * yps, o);
            ssum[0] += o[0]; ssq[0] += o[0] * o[0];
            ssum[1] += o[1]; ssq[1] += o[1] * o[1];
          }
        }
      }
    }
    sn = n; sty = ty; stx = tx;
    tx = ntx; ty = nty; n = nn;
    rowmin = nrowmin; base = nbase;
  }
  if (DW_FWD_STAGE) {
    __syncthreads();
    if (sn >= 0) store_y(sn, sty, stx);
  }

  if (stats) {
#pragma unroll
    for (int o = 32; o >= C2; o >>= 1) {
      ssum[0] += __shfl_xor(ssum[0], o, 64); ssum[1] += __shfl_xor(ssum[1], o, 64);
      ssq[0] += __shfl_xor(ssq[0], o, 64); ssq[1] += __shfl_xor(ssq[1], o, 64);
    }
    // no atomics anywhere: lanes 0..C2-1 of every wave hold that wave's sums, the four waves are added in order, and the
    // workgroup owns row `worker` of the partial-row buffer (bit-reproducible statistics)
    __syncthreads();
    if ((tid & 63) < C2) {
      float* sw = s_st + (tid >> 6) * 2 * CB;
      sw[2 * c2] = ssum[0];
      sw[2 * c2 + 1] = ssum[1];
      sw[CB + 2 * c2] = ssq[0];
      sw[CB + 2 * c2 + 1] = ssq[1];
    }
    __syncthreads();
    for (int i = tid; i < 2 * CB; i += 256) {
      const int pl = i / CB, c = c_base + i % CB;
      if (c < g.C) {
        const float v = ((s_st[i] + s_st[2 * CB + i]) + s_st[4 * CB + i]) + s_st[6 * CB + i];
        const long elem = (long)pl * stat_ld + c;
        stats[(long)worker * 2 * stat_ld + elem] = v;
        stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, elem);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Tiles are in INPUT space (TH x TW input pixels); the LDS tile holds dYraw over the output window those pixels touch.
// Work item = (channel pair, input row, strip of SW input pixels).  For stride 2 only taps of matching parity contribute:
// per (t, kx) that is a compile-time fact (tile and strip origins are even), per ky it is uniform for a row.
template <typename T, int K, int S, int SW, int CB, int TM, int AM>   // TM: largest tile edge (14, or 7 for the 7x7 maps); AM as forward
// (launch bounds for 3 resident workgroups, i.e. <= 168 VGPRs, make k = 5 spill 136 bytes and run 2.3x slower: measured, dropped)
__global__ __launch_bounds__(256, (K <= 5 ? DW_BWD_MINB5 : 2)) void k_dwconv_bwd(const T* __restrict__ gup, int ldg, long gss, const T* __restrict__ yraw, int ldyr,
                                                    long yrss, const float* __restrict__ c1, const float* __restrict__ c2p,
                                                    const float* __restrict__ c3, const T* __restrict__ x, int ldx, long xss,
                                                    const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                    int in_relu, const float* __restrict__ w, int ldw, T* __restrict__ h, int ldh,
                                                    long hss, float* __restrict__ dwp /*[nworkers][C][K*K] partial weight gradients*/,
                                                    float* __restrict__ stats, int stat_ld, int stat_rows, DwGeom g) {
  constexpr int P = (K - 1) / 2;
  constexpr int KK = K * K;
  constexpr int RELMIN = fdiv(-P, S);            // first output column (relative to strip origin / S) a strip touches
  constexpr int RELMAX = fdiv(SW - 1 + P, S);
  constexpr int DWN = RELMAX - RELMIN + 1;       // dY columns per strip
  static_assert(SW % S == 0, "strip origins must stay multiples of the stride");
  constexpr int LMAXB = fdiv(TM - 1 + P, S) - cdiv(P - (K - 1), S) + 1;  // dY window of a TM-pixel input tile
  constexpr int PF = (LMAXB * (fdiv(TM - 1 + P, S) - fdiv(-P, S) + 1) * (CB / 8) + 255) / 256;
  static_assert(PF <= 32, "prefetch mask is 32 bits");
  static_assert((TM * TM * (CB / 8) + 255) / 256 <= 32, "x prefetch mask is 32 bits");
  constexpr bool DMA = DW_DMA && sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // raw staging of the next tile's dY / yraw pieces (lane-linear: piece i of thread t at (i*256 + t) * 16 bytes); first in LDS
  T* s_rawg = reinterpret_cast<T*>(smem);                        // [PF][256][8]
  T* s_rawy = s_rawg + (DMA ? PF * 256 * 8 : 0);                 // [PF][256][8]
  float* s_dy = reinterpret_cast<float*>(s_rawy + (DMA ? PF * 256 * 8 : 0));   // [LH][RP]
  float* s_w = s_dy + g.LH * g.RP;           // [KK][CB]
  float* s_red = s_w + KK * CB;              // [CB][KK + 2]
  float* s_cf = s_red + CB * (KK + 2);       // [3][CB] BN-backward coefficients of the slab (c1 = 1, c2 = c3 = 0 without them)
  T* s_x = reinterpret_cast<T*>(s_cf + 3 * CB);   // [TH*TW][CB] raw input pixels of the tile (staged with 16-byte loads)
  T* s_h = s_x + TM * TM * CB;                    // [TH*TW][CB] the tile's input gradient, written out with 16-byte stores

  const int tid = threadIdx.x;
  constexpr int C2 = CB / 2, CG = CB / 8;
  const int b_xcd = blockIdx.x & 7, b_local = blockIdx.x >> 3;   // XCD-aware decode for the plain layout, see k_dwconv_fwd
  const int slab = g.xcd ? b_local % g.nslabs : (int)(blockIdx.x % g.nslabs);
  const int worker = g.xcd ? (b_local / g.nslabs) * 8 + b_xcd : (int)(blockIdx.x / g.nslabs);
  if (worker >= g.nworkers) return;
  const int c_base = slab * CB;
  const int cpad = (g.C + 7) & ~7;

  for (int i = tid; i < KK * CB; i += 256) {
    const int t = i / CB, c = i % CB;
    s_w[i] = (c_base + c < g.C) ? w[(long)t * ldw + c_base + c] : 0.f;
  }
  for (int i = tid; i < CB * (KK + 2); i += 256) s_red[i] = 0.f;
  for (int i = tid; i < 3 * CB; i += 256) {
    const int v = i / CB, c = c_base + i % CB;
    const float* src = (v == 0) ? c1 : (v == 1 ? c2p : c3);
    s_cf[i] = (c1 && src && c < cpad) ? src[c] : (v == 0 ? 1.f : 0.f);
  }

  const int cg = tid % CG;
  const bool cg_ok = c_base + cg * 8 < cpad;

  const int cc2 = tid % C2;
  const int ch = c_base + 2 * cc2;
  const bool ch_ok = ch < cpad;
  float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
  if (in_scale && ch_ok) { VecIO<float, 2>::load(in_scale + ch, sc); VecIO<float, 2>::load(in_shift + ch, sh); }
  const int nstrips = g.TW / SW;
  const int nitems = C2 * g.TH * nstrips;

  f32x2 dwa[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) dwa[t] = f32x2{0.f, 0.f};
  float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};

  const int ntiles = g.N * g.tiles_y * g.tiles_x;
  const int npix = g.LH * g.LW;
  int p_iy[PF], p_ix[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int pidx = tid / CG + i * (256 / CG);
    p_iy[i] = pidx < npix ? pidx / g.LW : -100000;
    p_ix[i] = pidx % g.LW;
  }
  Raw8<T> pfg[DMA ? 1 : PF], pfy[DMA ? 1 : PF];
  unsigned pfmask = 0;
  const unsigned rawg_base = lds_addr(s_rawg) + (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6) * 64u * 16u;
  const unsigned rawy_base = lds_addr(s_rawy) + (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6) * 64u * 16u;
  auto issue = [&](int n, int ty, int tx, int rowmin) {
    const int hob = cdiv(ty * g.TH + P - (K - 1), S), wob = (tx * g.TW) / S + RELMIN;
    const long gps = pix_stride(ldg, gss), yps = pix_stride(ldyr, yrss);
    const int cgc = cg_ok ? c_base + cg * 8 : 0;   // a channel group beyond C reads group 0 (masked), never beyond the tensor
    const T* gn = gup + (long)n * g.Ho * g.Wo * gps + chan_base(cgc, gss);
    const T* yn = yraw ? yraw + (long)n * g.Ho * g.Wo * yps + chan_base(cgc, yrss) : nullptr;
    pfmask = 0;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int ho = hob + p_iy[i], wo = wob + p_ix[i];
      const bool ok = DW_EXP != 8 && DW_EXP != 11 && cg_ok && p_iy[i] >= rowmin && ho >= 0 && ho < g.Ho && wo >= 0 && wo < g.Wo;
      if constexpr (DMA) {
        if (ok) {
          const long off = (long)ho * g.Wo + wo;
          lds_dma16(gn + off * gps, rawg_base + (unsigned)i * 256u * 16u);
          if (yn && DW_EXP != 9) lds_dma16(yn + off * yps, rawy_base + (unsigned)i * 256u * 16u);
        }
      } else {
        // Branch-free (round 3): a piece outside the image / the channel range reads the first piece of the image instead and is
        // masked at the commit.  With the loads under per-lane branches hipcc cannot prove at the loop back-edge that they were
        // waited for and puts an s_waitcnt vmcnt(0) in front of the next tile's loads, which also waits for the h stores issued
        // just before (csrc/dwconv_cw.hip, same finding: 20-30 % of the kernel).
        const long off = ok ? (long)ho * g.Wo + wo : 0;
        pfg[i].load(gn + off * gps);
        if (yn && DW_EXP != 9) pfy[i].load(yn + off * yps);
      }
      pfmask |= ok ? 1u << i : 0u;
    }
  };
  auto commit = [&](int rowmin, int base) {
    if constexpr (DMA) lds_dma_wait();   // this thread's own pieces have landed (no other thread reads them)
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (p_iy[i] >= rowmin) {
        float v[8];
        const bool ok = (pfmask >> i) & 1u;
        Raw8<T> rg, ry;
        if constexpr (DMA) {
          rg.zero(); ry.zero();
          if (ok) {
            rg.load(s_rawg + (i * 256 + tid) * 8);
            if (yraw) ry.load(s_rawy + (i * 256 + tid) * 8);
          }
        } else {
          rg = pfg[i]; ry = pfy[i];
        }
        float q1[8], q2[8], q3[8];   // re-read from LDS per tile: keeps 24 registers free during the FMA phase
        VecIO<float, 8>::load(s_cf + cg * 8, q1);
        VecIO<float, 8>::load(s_cf + CB + cg * 8, q2);
        VecIO<float, 8>::load(s_cf + 2 * CB + cg * 8, q3);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = q1[e] * rg.get(e);
          if (yraw) a += q2[e] * ry.get(e) + q3[e];
          v[e] = ok ? a : 0.f;
        }
        int slot = p_iy[i] + base;
        if (slot >= g.LH) slot -= g.LH;
        float* d = s_dy + slot * g.RP + p_ix[i] * CB + cg * 8;
        *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
    }
  };
  constexpr int NIT = (C2 * TM * (TM / SW) + 255) / 256;
  int it_r[NIT], it_j[NIT];
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int it = tid + q * 256;
    const int rs = it / C2;
    it_r[q] = it < nitems ? rs / nstrips : -1;
    it_j[q] = rs % nstrips;
  }
  // The tile's input pixels x and its result h move between HBM and LDS in 16-byte pieces (pixel, 8 channels); the work
  // items read / write their channel pairs in LDS.  Measured (cold caches, 56x56x144, k = 3): with 4-byte-per-lane global
  // accesses the x loads alone cost 31 % of the kernel (65 % for stride 2) and the h stores 14 %.  The strip's raw pixels
  // stay in registers from the weight-gradient operand to the ReLU mask / statistics of the epilogue.
  constexpr int XP = (TM * TM * CG + 255) / 256;
  int xp_r[XP], xp_c[XP];
#pragma unroll
  for (int p = 0; p < XP; ++p) {
    const int pix = (tid + 256 * p) / CG;
    xp_r[p] = pix < g.TH * g.TW ? pix / g.TW : -100000;
    xp_c[p] = pix % g.TW;
  }
  Raw8<T> xr[XP];
  unsigned xmask = 0;
  auto issue_x = [&](int an, int aty, int atx) {   // branch-free, see issue()
    xmask = 0;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int hi = aty * g.TH + xp_r[p], wi = atx * g.TW + xp_c[p];
      const bool ok = DW_EXP != 7 && DW_EXP != 11 && cg_ok && hi >= 0 && hi < g.H && wi < g.W;
      const long pix = ok ? ((long)an * g.H + hi) * g.W + wi : (long)an * g.H * g.W;
      xr[p].load(x + pix * pix_stride(ldx, xss) + chan_base(cg_ok ? c_base + cg * 8 : 0, xss));
      xmask |= ok ? 1u << p : 0u;
    }
  };
  auto commit_x = [&]() {
#pragma unroll
    for (int p = 0; p < XP; ++p)
      if (xp_r[p] >= 0) {
        Raw8<T> v = xr[p];
        if (!((xmask >> p) & 1u)) v.zero();
        v.store(s_x + (xp_r[p] * g.TW + xp_c[p]) * CB + cg * 8);
      }
  };
  auto store_h = [&](int an, int aty, int atx) {
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int hi = aty * g.TH + xp_r[p], wi = atx * g.TW + xp_c[p];
      if (DW_EXP != 6 && DW_EXP != 11 && cg_ok && hi >= 0 && hi < g.H && wi < g.W) {
        Raw8<T> v;
        v.load(s_h + (xp_r[p] * g.TW + xp_c[p]) * CB + cg * 8);
        v.store(h + (((long)an * g.H + hi) * g.W + wi) * pix_stride(ldh, hss) + chan_base(c_base + cg * 8, hss));
      }
    }
  };
  Raw2<T> xq[NIT][SW];
#if DW_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif

  const int t_beg = (int)((long)worker * ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * ntiles / g.nworkers);
  int tile = t_beg;   // contiguous tile range per worker (halo reuse in the XCD's L2), see k_dwconv_fwd
  int ty = tile % g.tiles_y, tx = (tile / g.tiles_y) % g.tiles_x, n = tile / (g.tiles_x * g.tiles_y);   // column-major, see k_dwconv_fwd
  auto advance = [&](int& an, int& aty, int& atx) {
    if (++aty == g.tiles_y) { aty = 0; if (++atx == g.tiles_x) { atx = 0; ++an; } }
  };
  auto hob_of = [&](int aty) { return cdiv(aty * g.TH + P - (K - 1), S); };
  int ntx = tx, nty = ty, nn = n;
  int rowmin = 0, base = 0, nrowmin = 0, nbase = 0;   // ring state of the dY tile, see k_dwconv_fwd
#ifndef DW_XPRE
#define DW_XPRE 1
#endif
  constexpr bool XPRE = DW_XPRE && (K < 7);   // x joins the one-tile-ahead prefetch where the registers allow it
  if (tile < t_end) { issue(n, ty, tx, 0); if (XPRE) issue_x(n, ty, tx); }
  int hn = -1, hty = 0, htx = 0;   // tile whose result is waiting in s_h
  for (; tile < t_end; ++tile) {
    const int hi0 = ty * g.TH, wi0 = tx * g.TW;                 // multiples of S (TH, TW even when S == 2)
    const int hob = cdiv(hi0 + P - (K - 1), S);                 // first output row held in LDS
    TMARK(6)
    if (!XPRE) issue_x(n, ty, tx);
    __syncthreads();   // previous tile fully consumed, its result complete in s_h
    TMARK(0)
    if constexpr (!DMA) {   // every prefetched register is read here on every path: nothing is pending at the next issue
#pragma unroll
      for (int i = 0; i < PF; ++i) { dw_touch(pfg[i]); dw_touch(pfy[i]); }
#pragma unroll
      for (int p = 0; p < XP; ++p) dw_touch(xr[p]);
    }
    if (hn >= 0) store_h(hn, hty, htx);
    commit(rowmin, base);
    commit_x();
    TMARK(1)
    __syncthreads();
    TMARK(2)
    ntx = tx; nty = ty; nn = n;
    advance(nn, nty, ntx);
    {
      const int vshift = hob_of(ty + 1) - hob;   // rows the dY window moves down to the tile below
      const bool below = DW_RING && vshift < g.LH && nn == n && ntx == tx && nty == ty + 1;
      nrowmin = below ? g.LH - vshift : 0;
      nbase = below ? base + vshift : 0;
      if (nbase >= g.LH) nbase -= g.LH;
    }
    if (tile + 1 < t_end) { issue(nn, nty, ntx, nrowmin); if (XPRE) issue_x(nn, nty, ntx); }
    TMARK(3)

#pragma unroll
    for (int q = 0; q < (DW_EXP == 5 ? 0 : NIT); ++q) {
      const int r = it_r[q], j = it_j[q];
      const int hi = hi0 + r;
      if (r < 0 || hi >= g.H || !ch_ok) continue;
      const int wis = wi0 + j * SW;  // first input column of this strip (multiple of S)
      // this strip's activated input pixels (for the weight gradient)
      f32x2 xa[SW];
      const int pix0 = r * g.TW + j * SW;   // first pixel of the strip inside the tile
#pragma unroll
      for (int t = 0; t < SW; ++t) xq[q][t].load(s_x + (pix0 + t) * CB + 2 * cc2);
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        const int wi = wis + t;
        xa[t] = f32x2{0.f, 0.f};
        if (wi < g.W) {
          const float v[2] = {xq[q][t].get(0), xq[q][t].get(1)};
          const float a0 = v[0] * sc[0] + sh[0], a1 = v[1] * sc[1] + sh[1];
          if constexpr (AM == ACT_RELU6) xa[t] = f32x2{fminf(fmaxf(a0, 0.f), 6.f), fminf(fmaxf(a1, 0.f), 6.f)};
          else if constexpr (AM == ACT_SWISH) xa[t] = f32x2{swish_f(a0), swish_f(a1)};
          else xa[t] = in_relu ? f32x2{fmaxf(a0, 0.f), fmaxf(a1, 0.f)} : f32x2{a0, a1};
        }
        asm volatile("" : "+v"(xa[t]));   // computed here, before the tap rows
      }
      asm volatile("" ::: "memory");
      f32x2 dx[SW];
#pragma unroll
      for (int t = 0; t < SW; ++t) dx[t] = f32x2{0.f, 0.f};

#pragma unroll
      for (int ky = 0; ky < (DW_EXP == 10 ? 0 : K); ++ky) {
        const int numr = hi + P - ky;
        if (S > 1 && pmod(numr, S) != 0) continue;
        const int ho = fdiv(numr, S);
        if (ho < 0 || ho >= g.Ho) continue;   // rows outside the image hold zeros anyway; skip the work
        int slot = ho - hob + base;
        if (slot >= g.LH) slot -= g.LH;
        const float* row = s_dy + slot * g.RP + ((wis - wi0) / S) * CB + 2 * cc2;
        f32x2 dy[DWN];
        {
          // single ds_read_b64 each (inline asm): hipcc merges neighbours into ds_read2_b64, which moves the same bytes in twice
          // the LDS cycles (MI355X_MICROARCH.md, LDS table); the wait and the fence make the results valid before their use
          const unsigned ra = lds_addr(row);
#pragma unroll
          for (int i = 0; i < DWN; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dy[i]) : "v"(ra), "i"(i * CB * 4));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const f32x2 wv = *reinterpret_cast<const f32x2*>(s_w + (ky * K + kx) * CB + 2 * cc2);
#pragma unroll
          for (int t = 0; t < SW; ++t) {
            const int num = t + P - kx;  // compile-time after unrolling
            if (pmod(num, S) != 0) continue;
            const int jj = fdiv(num, S) - RELMIN;
            dx[t] += dy[jj] * wv;
            dwa[ky * K + kx] += xa[t] * dy[jj];
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep one dY row live at a time (the loop is unrolled for static dwa indices)
      }

      TMARK(4)
      // epilogue: ReLU mask of the producer, rounding, statistics; the result goes to LDS and leaves with the next tile.  The raw
      // pixels are read again from LDS rather than kept in registers across the tap rows (k = 7 spilled: round 3).
#pragma unroll
      for (int t = 0; t < SW; ++t) xq[q][t].load(s_x + (pix0 + t) * CB + 2 * cc2);
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        const int wi = wis + t;
        if (wi < g.W) {
          const float xv[2] = {xq[q][t].get(0), xq[q][t].get(1)};
          float o[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float a = xv[c] * sc[c] + sh[c];
            float v;
            if constexpr (AM == ACT_RELU6) v = (a > 0.f && a < 6.f) ? dx[t][c] : 0.f;
            else if constexpr (AM == ACT_SWISH) v = dx[t][c] * swish_grad(a);
            else v = (in_relu && !(a > 0.f)) ? 0.f : dx[t][c];
            v = (ch + c < g.C) ? to_f32(from_f32<T>(v)) : 0.f;
            o[c] = v;
            s0[c] += v;
            s1[c] += v * xv[c];
          }
          VecIO<T, 2>::store(s_h + (pix0 + t) * CB + 2 * cc2, o);
        }
      }
      TMARK(5)
    }
    hn = n; hty = ty; htx = tx;
    tx = ntx; ty = nty; n = nn;
    rowmin = nrowmin; base = nbase;
  }
  __syncthreads();
  if (hn >= 0) store_h(hn, hty, htx);
#if DW_TIMING
  if ((tid & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) atomicAdd(&g_dw_timing[i], tacc[i]);
    atomicAdd(&g_dw_timing[7], (unsigned long long)(t_end - t_beg));
  }
#endif

  // block reduction of the weight gradient and the statistics, one flush per workgroup: lanes l, l+C2, l+2*C2, ... of a
  // wave hold the same channel pair -> butterfly over those first, then the first C2 lanes of each wave add their values in LDS
  {
    const int lane = tid & 63;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
#pragma unroll
      for (int o = 32; o >= C2; o >>= 1) {
        dwa[t][0] += __shfl_xor(dwa[t][0], o, 64);
        dwa[t][1] += __shfl_xor(dwa[t][1], o, 64);
      }
    }
#pragma unroll
    for (int o = 32; o >= C2; o >>= 1) {
      s0[0] += __shfl_xor(s0[0], o, 64); s0[1] += __shfl_xor(s0[1], o, 64);
      s1[0] += __shfl_xor(s1[0], o, 64); s1[1] += __shfl_xor(s1[1], o, 64);
    }
    // the four waves add their values one after the other (plain read-modify-write, distinct addresses within a wave):
    // a fixed order instead of LDS atomics, so the result is bit-reproducible
    for (int wv = 0; wv < 4; ++wv) {
      if ((tid >> 6) == wv && lane < C2) {
        float* r0 = &s_red[(2 * cc2) * (KK + 2)];
        float* r1 = &s_red[(2 * cc2 + 1) * (KK + 2)];
#pragma unroll
        for (int t = 0; t < KK; ++t) {
          r0[t] += dwa[t][0];
          r1[t] += dwa[t][1];
        }
        r0[KK] += s0[0]; r0[KK + 1] += s1[0];
        r1[KK] += s0[1]; r1[KK + 1] += s1[1];
      }
      __syncthreads();
    }
  }
  // flush: this workgroup owns row `worker` of the weight-gradient partials and of the statistics (plain stores; the
  // partials are summed by reduce_parts, the statistics by the BatchNorm finalize kernel, both in a fixed order)
  for (int i = tid; i < CB * (KK + 2); i += 256) {
    const int cl = i / (KK + 2), t = i % (KK + 2);
    const int c = c_base + cl;
    if (c >= g.C) continue;
    const float v = s_red[i];
    if (DW_EXP == 4) continue;
    if (t < KK) {
      if (dwp) dwp[((long)worker * g.C + c) * KK + t] = v;
    } else if (stats) {
      const long elem = (long)(t - KK) * stat_ld + c;
      stats[(long)worker * 2 * stat_ld + elem] = v;
      stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, elem);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int slab_width(int preferred, int cpad) {
  int need = cpad <= 8 ? 8 : (cpad <= 16 ? 16 : (cpad <= 32 ? 32 : 64));
  return preferred < need ? preferred : need;
}

// Tile configuration.  rows/cols: extent of the tiled space (output pixels forward, input pixels backward).
// (Measured and dropped: 7 x 28 tiles walked column-major, which keep the vertical halo in L2 and cut the HBM-side re-fetch,
// run exactly as fast -- the kernels are bound by the number of L1/L2-side requests, halo included, not by HBM-side bytes.)
static void pick_tiles(DwGeom& g, int rows, int cols, int sw, int cb, int even) {
  g.CB = cb;
  g.TH = rows < 14 ? rows : 14;
  g.TW = cols < 14 ? cols : 14;
  g.TW = (g.TW + sw - 1) / sw * sw;  // whole strips
  if (even && (g.TH % 2)) g.TH += 1;
  g.tiles_y = (rows + g.TH - 1) / g.TH;
  g.tiles_x = (cols + g.TW - 1) / g.TW;
}

// Persistent grid: exactly as many workgroups as are resident at once (a partial second round of workgroups costs up to 2x:
// a workgroup that does not fit waits for a whole worker lifetime).  Workgroup b runs on XCD b % 8, so what has to fit is the
// PER-XCD count.  With the XCD-aware decode (plain layout) worker w lives on XCD w % 8: ceil(workers / 8) * nslabs workgroups on
// (CUs / 8) * per_cu slots, i.e. whole multiples of 8 workers.  (Round 2 found workers = total_slots / nslabs here, e.g. 17
// workers x 30 slabs at 14x14x480: XCD 0 got 3 x 30 = 90 workgroups for 64 slots and the kernel ran two rounds; fixing it took
// the depthwise kernels from 25.4 to 20.7 ms per step.)  In plain order (slab-major layout) any count <= total slots is level.
static void set_workers(DwGeom& g, int nslabs, int per_cu, int cap, long max_workers, bool xcd_decode) {
  const long ntiles = (long)g.N * g.tiles_y * g.tiles_x;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > cap) per_cu = cap;
  constexpr int level_env = 2;   // A/B: 0 round-1 rule, 1 aligned
  const bool aligned = level_env == 1 || (level_env == 2 && xcd_decode);
  g.xcd = (level_env == 2 && !xcd_decode) ? 0 : 1;
  const long slots_xcd = (long)(num_cus() / 8) * per_cu;
  long want = aligned ? (slots_xcd / nslabs) * 8 : ((long)num_cus() * per_cu) / nslabs;
  if (want < 8) want = ((long)num_cus() * per_cu) / nslabs;   // more slabs than slots of an XCD: several rounds either way
  constexpr int share = 1;   // experiment: one of `share` concurrent launches
  if (share > 1) want = want / share > 0 ? want / share : 1;
  static const long max_env = getenv("ATOMNAS_DW_MAX_WORKERS") ? atol(getenv("ATOMNAS_DW_MAX_WORKERS")) : 0;   // tests: force long tile walks
  if (max_env > 0 && want > max_env) want = max_env;
  if (max_workers > 0 && want > max_workers) want = max_workers;   // every worker owns one partial row (statistics, weight gradient)
  if (want > ntiles) want = ntiles;
  if (aligned && want > 8) want = want / 8 * 8;
  if (want < 1) want = 1;
  g.nworkers = (int)want;
  g.nslabs = nslabs;
}
// grid: XCD-aware decode = whole worker groups per XCD (surplus workgroups exit at once); plain order = workers x slabs
static inline unsigned dw_grid(const DwGeom& g) {
  return g.xcd ? (unsigned)((g.nworkers + 7) / 8 * 8 * g.nslabs) : (unsigned)(g.nworkers * g.nslabs);
}

template <typename T, int K, int S>
static int launch_fwd(const void* x, int ldx, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y,
                      int ldy, long yss, float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, hipStream_t st) {
  constexpr int P = (K - 1) / 2;
  DwGeom g;
  g.N = N; g.H = H; g.W = W; g.C = C;
  g.Ho = (H + 2 * P - K) / S + 1; g.Wo = (W + 2 * P - K) / S + 1;
  const int cpad = (C + 7) / 8 * 8;
  // 14x14 output tiles of 16 channels (several workgroups per CU); small maps take the whole image and 64 channels
  const int sw = 7;
  constexpr int cb_env = 0;
  // default 16; deviations measured in situ on the supernet's own shapes (bs 256 step, tools/bringup.py DETAIL=1 +
  // tools/cmpdetail.py, re-done after the XCD-level fix of set_workers): 8-channel slabs for the stride-2 layers with few channels
  // per pixel, 32 for C <= 32
  // r03: for slab-major tensors whole slabs win on every stride-2 shape of the step since the prefetch became branch-free (8-channel
  // workgroups read half of every 32-byte slab row; tools/dwbench.py fwd, ATOMNAS_DW_FWD_CB=16 vs the rule: 1.55 -> 1.36 ms)
  int cb_rule = 16;
  // One rule for both storage types (round 5).  Rounds 3-4 kept 8-channel slabs for fp32 because regrouping the statistics partials
  // moved tests/golden/checkpoint_ref.pt's element-wise comparison across a ReLU-mask flip; the fixture's resume batch had a
  // pre-activation ON zero (|pre| / rms 7.8e-9).  The fixture now resumes on a wide-margin batch (tools/make_golden.py: 1.75e-6), so
  // the comparison no longer depends on the summation order.
  const bool whole_slabs = xss != 0;
  if (S == 2 && C <= 96 && !whole_slabs) cb_rule = 8;
  else if (S == 2 && K >= 5 && H <= 56 && !whole_slabs) cb_rule = 8;
  else if (S == 1 && C <= 32) cb_rule = 32;
  const int cb = slab_width((g.Wo <= 7 && g.Ho <= 7) ? 64 : (cb_env ? cb_env : cb_rule), cpad);
  pick_tiles(g, g.Ho, g.Wo, sw, cb, 0);
  ATOMNAS_REQUIRE(cb <= 32 || (g.TH <= 7 && g.TW <= 7), "dwconv_fwd: internal tile configuration error");
  g.LH = (g.TH - 1) * S + K;
  g.LW = (g.TW - 1) * S + K;
  g.RP = lds_pitch(g.LW, cb);
  const int nslabs = (cpad + cb - 1) / cb;
  const size_t lds = ((size_t)g.LH * g.RP + (size_t)K * K * cb + 8 * cb) * sizeof(float) + (DW_FWD_STAGE ? (size_t)g.TH * g.TW * cb * sizeof(T) : 0);
  ATOMNAS_REQUIRE(lds <= max_lds_bytes(), "dwconv_fwd: tile does not fit in LDS (%zu bytes)", lds);
  constexpr int cap_env = 0;
  const int cap = cap_env ? cap_env : 8;
#define FWD_CASE(CBV, TMV)                                                                                                   \
  {                                                                                                                      \
    auto kern = (relu == ACT_RELU6) ? k_dwconv_fwd<T, K, S, 7, CBV, TMV, ACT_RELU6>                                    \
                                    : (relu == ACT_SWISH ? k_dwconv_fwd<T, K, S, 7, CBV, TMV, ACT_SWISH> : k_dwconv_fwd<T, K, S, 7, CBV, TMV, 0>);                                                                           \
    set_workers(g, nslabs, resident_per_cu(kern, 256, lds), cap, stats ? stat_rows : 0, xss == 0);                                      \
    dim3 grid(dw_grid(g));                                                                                      \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const T*)x, ldx, xss, sc, sh, relu, w, ldw, (T*)y, ldy, yss, stats, stat_ld, stat_rows, g); \
  }
  const bool small = g.TH <= 7 && g.TW <= 7;
  if (cb == 8) { if (small) FWD_CASE(8, 7) else FWD_CASE(8, 14) }
  else if (cb == 16) { if (small) FWD_CASE(16, 7) else FWD_CASE(16, 14) }
  else if (cb == 32) { if (small) FWD_CASE(32, 7) else FWD_CASE(32, 14) }
  else FWD_CASE(64, 7)
#undef FWD_CASE
  return check_launch("dwconv_fwd");
}

template <typename T, int K, int S, int SW>
static int launch_bwd_sw(const void* gup, int ldg, long gss, const void* yraw, int ldyr, long yrss, const float* c1, const float* c2,
                         const float* c3, const void* x, int ldx, long xss, const float* sc, const float* sh, int relu, const float* w,
                         int ldw, void* h, int ldh, long hss, float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C,
                         hipStream_t st) {
  constexpr int P = (K - 1) / 2;
  DwGeom g;
  g.N = N; g.H = H; g.W = W; g.C = C;
  g.Ho = (H + 2 * P - K) / S + 1; g.Wo = (W + 2 * P - K) / S + 1;
  const int cpad = (C + 7) / 8 * 8;
  // measured (tools/dwbench.py): 32-channel slabs win for stride 2 and for 7x7 maps with k <= 5, 16-channel slabs elsewhere
  // (k = 7 with 32 channels spills its 98 weight-gradient accumulators)
  constexpr int cb_env = 0;
  // in-situ deviations (same sweep as the forward, re-done after the XCD-level fix of set_workers): k = 5 at 28x28 and 14x14
  // prefers 32
  int cb_rule = (S == 2 || (H <= 7 && W <= 7 && K <= 5)) ? 32 : 16;
  if (S == 1 && H == 28 && K == 5) cb_rule = 32;
  else if (S == 1 && H == 14 && K == 5) cb_rule = 32;
  // 7x7 maps: the whole image is one tile of 7 rows x 1 strip, so only wide slabs fill the 256 threads (64 channels: 224 work
  // items; 16 channels: 56).  The prefetch registers are sized for the 7-pixel tile there (template parameter TM).
  const bool small = (S == 1 && H <= 7 && W <= 7);
  if (small) cb_rule = (K == 7) ? DW_SMALL_CB7 : 64;
  if (S == 1 && SW == 14) cb_rule = 32;   // one 14-pixel strip per row: 16 channel pairs x 14 rows = 224 work items
  const int cb = slab_width((S == 1 && SW == 14) ? cb_rule : (cb_env ? cb_env : cb_rule), cpad);
  pick_tiles(g, H, W, SW, cb, S == 2);
  // output window of an input tile: rows ceil((hi0+P-K+1)/S) .. floor((hi0+TH-1+P)/S)
  g.LH = fdiv(g.TH - 1 + P, S) - cdiv(P - (K - 1), S) + 1;
  g.LW = fdiv(g.TW - 1 + P, S) - fdiv(-P, S) + 1;
  g.RP = lds_pitch(g.LW, cb);
  const int nslabs = (cpad + cb - 1) / cb;
  // + raw LDS-DMA staging of the dY / yraw prefetch (bf16): 2 streams x PF pieces x 256 threads x 16 bytes
  const int tm = small ? 7 : 14;
  const int lmaxb = fdiv(tm - 1 + P, S) - cdiv(P - (K - 1), S) + 1, lmaxw = fdiv(tm - 1 + P, S) - fdiv(-P, S) + 1;
  const int pf = (lmaxb * lmaxw * (cb / 8) + 255) / 256;
  const size_t lds = ((size_t)g.LH * g.RP + (size_t)K * K * cb + (size_t)cb * (K * K + 2) + 3 * (size_t)cb) * sizeof(float) +
                     (size_t)2 * tm * tm * cb * sizeof(T) + ((DW_DMA && sizeof(T) == 2) ? (size_t)2 * pf * 256 * 16 : 0);
  ATOMNAS_REQUIRE(lds <= max_lds_bytes(), "dwconv_bwd: tile does not fit in LDS (%zu bytes)", lds);
  constexpr int cap_env2 = 0;
  const int cap = cap_env2 ? cap_env2 : 8;
#define BWD_CASE(CBV, TMV)                                                                                                \
  {                                                                                                                       \
    auto kern = (relu == ACT_RELU6) ? k_dwconv_bwd<T, K, S, SW, CBV, TMV, ACT_RELU6>                                   \
                                    : (relu == ACT_SWISH ? k_dwconv_bwd<T, K, S, SW, CBV, TMV, ACT_SWISH> : k_dwconv_bwd<T, K, S, SW, CBV, TMV, 0>);                                                                           \
    set_workers(g, nslabs, resident_per_cu(kern, 256, lds), cap, (stats || dw) ? part_rows : 0, xss == 0 || gss == 0);                               \
    dim3 grid(dw_grid(g));                                                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const T*)gup, ldg, gss, (const T*)yraw, ldyr, yrss, c1, c2, c3, (const T*)x, \
                       ldx, xss, sc, sh, relu, w, ldw, (T*)h, ldh, hss, dw ? dw_ws : nullptr, stats, stat_ld, part_rows, g); \
  }
  if constexpr (S == 1 && SW == 14) {
    ATOMNAS_REQUIRE(cb == 32 && !small, "dwconv_bwd: wide strips need 32-channel slabs and tiles of 14 pixels");
    BWD_CASE(32, 14)
  } else if constexpr (S == 1) {
    if (small) {
      if (cb == 8) BWD_CASE(8, 7) else if (cb == 16) BWD_CASE(16, 7) else if (cb == 32) BWD_CASE(32, 7) else BWD_CASE(64, 7)
    } else {
      if (cb == 8) BWD_CASE(8, 14) else if (cb == 16) BWD_CASE(16, 14) else BWD_CASE(32, 14)
    }
  } else {
    if (cb == 8) BWD_CASE(8, 14) else if (cb == 16) BWD_CASE(16, 14) else BWD_CASE(32, 14)
  }
#undef BWD_CASE
  if (int rc = check_launch("dwconv_bwd")) return rc;
  // dw[c][t] += sum over workers of the partials, in worker order
  if (dw) return reduce_parts(dw_ws, (long)C * K * K, g.nworkers, (long)C * K * K, dw, C * K * K, 0, 1, st);
  return 0;
}

template <typename T, int K, int S>
static int launch_bwd(const void* gup, int ldg, long gss, const void* yraw, int ldyr, long yrss, const float* c1, const float* c2,
                      const float* c3, const void* x, int ldx, long xss, const float* sc, const float* sh, int relu, const float* w,
                      int ldw, void* h, int ldh, long hss, float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C,
                      hipStream_t st) {
  // experiment (ATOMNAS_DW_BWD_SW14=k-mask, bit 0: k = 3, bit 1: k = 5): 14-pixel strips for stride 1 -- 27 % fewer LDS reads per FMA
  if constexpr (sizeof(T) == 2 && S == 1 && K <= 5) {
    constexpr int wide_env = 0;
    if (((wide_env >> (K == 3 ? 0 : 1)) & 1) && H > 7 && W >= 14 && C >= 32)
      return launch_bwd_sw<T, K, S, 14>(gup, ldg, gss, yraw, ldyr, yrss, c1, c2, c3, x, ldx, xss, sc, sh, relu, w, ldw, h, ldh, hss, dw, stats,
                                         stat_ld, part_rows, dw_ws, N, H, W, C, st);
  }
  return launch_bwd_sw<T, K, S, (S == 2) ? 14 : 7>(gup, ldg, gss, yraw, ldyr, yrss, c1, c2, c3, x, ldx, xss, sc, sh, relu, w, ldw, h, ldh, hss, dw,
                                                    stats, stat_ld, part_rows, dw_ws, N, H, W, C, st);
}

#if DW_TIMING
}  // namespace atomnas
extern "C" int atomnas_debug_dw_timing(unsigned long long* out8, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(atomnas::g_dw_timing), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(atomnas::g_dw_timing), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
namespace atomnas {
#endif

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

static inline bool lay_ok(int ld, long ss, int cpad, long rows) {
  return ss ? (ss >= rows * 16) : (ld >= cpad && ld % 8 == 0);
}

extern "C" int atomnas_dwconv_fwd(const void* x, int ldx, long x_ss, const float* in_scale, const float* in_shift, int in_relu,
                                  const float* w, int ldw, void* y, int ldy, long y_ss, float* stats, int stat_ld, int stat_rows, int N, int H, int W,
                                  int C, int k, int stride, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && w && y, "dwconv_fwd: null pointer");
  ATOMNAS_REQUIRE((k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2), "dwconv_fwd: unsupported k=%d stride=%d", k, stride);
  ATOMNAS_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv_fwd: bad dtype %d", dtype);
  ATOMNAS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "dwconv_fwd: empty shape");
  const int cpad = (C + 7) / 8 * 8;
  {
    const int P = (k - 1) / 2;
    const long mi = (long)N * H * W, mo = (long)N * ((H + 2 * P - k) / stride + 1) * ((W + 2 * P - k) / stride + 1);
    ATOMNAS_REQUIRE(lay_ok(ldx, x_ss, cpad, mi) && lay_ok(ldy, y_ss, cpad, mo) && ldw >= C,
                    "dwconv_fwd: bad pitch / slab stride (C=%d ldx=%d ldy=%d ldw=%d)", C, ldx, ldy, ldw);
  }
  ATOMNAS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dwconv_fwd: scale/shift must come together");
  ATOMNAS_REQUIRE(!stats || (stat_ld >= C && stat_rows > 0), "dwconv_fwd: statistics pitch %d < C=%d or stat_rows=%d", stat_ld, C, stat_rows);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    int rc = dwconv_mm_fwd(x, x_ss, in_scale, in_shift, in_relu, w, ldw, y, y_ss, stats, stat_ld, stat_rows, N, H, W, C, k, dtype, st);
    if (rc >= 0) return rc;
    rc = dwconv_cw_fwd(x, x_ss, in_scale, in_shift, in_relu, w, ldw, y, y_ss, stats, stat_ld, stat_rows, N, H, W, C, k, dtype, st);
    if (rc >= 0) return rc;
  } else {
    const int rc = dwconv_mm2_fwd(x, x_ss, in_scale, in_shift, in_relu, w, ldw, y, y_ss, stats, stat_ld, stat_rows, N, H, W, C, k, dtype, st);
    if (rc >= 0) return rc;
  }
  DW_DISPATCH(launch_fwd, x, ldx, x_ss, in_scale, in_shift, in_relu, w, ldw, y, ldy, y_ss, stats, stat_ld, stat_rows, N, H, W, C, st);
  return 1;
}

extern "C" int atomnas_dwconv_bwd(const void* g, int ldg, long g_ss, const void* yraw, int ldyr, long yraw_ss, const float* c1,
                                  const float* c2, const float* c3, const void* x, int ldx, long x_ss, const float* in_scale,
                                  const float* in_shift, int in_relu, const float* w, int ldw, void* h, int ldh, long h_ss, float* dw, float* stats, int stat_ld,
                                  int part_rows, float* dw_ws, int N, int H, int W, int C, int k, int stride, int dtype,
                                  void* stream) {
  ATOMNAS_REQUIRE(g && x && w && h, "dwconv_bwd: null pointer");
  ATOMNAS_REQUIRE((k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2), "dwconv_bwd: unsupported k=%d stride=%d", k, stride);
  ATOMNAS_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv_bwd: bad dtype %d", dtype);
  ATOMNAS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "dwconv_bwd: empty shape");
  const int cpad = (C + 7) / 8 * 8;
  const int Pk = (k - 1) / 2;
  const long mi = (long)N * H * W, mo = (long)N * ((H + 2 * Pk - k) / stride + 1) * ((W + 2 * Pk - k) / stride + 1);
  ATOMNAS_REQUIRE(lay_ok(ldx, x_ss, cpad, mi) && lay_ok(ldg, g_ss, cpad, mo) && lay_ok(ldh, h_ss, cpad, mi) && ldw >= C,
                  "dwconv_bwd: bad pitch / slab stride");
  ATOMNAS_REQUIRE(!yraw || (c1 && c2 && c3 && lay_ok(ldyr, yraw_ss, cpad, mo)), "dwconv_bwd: yraw needs c1,c2,c3 and a valid pitch");
  ATOMNAS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dwconv_bwd: scale/shift must come together");
  ATOMNAS_REQUIRE(!stats || stat_ld >= C, "dwconv_bwd: statistics pitch %d < C=%d", stat_ld, C);
  ATOMNAS_REQUIRE(!(stats || dw) || part_rows > 0, "dwconv_bwd: part_rows must be positive");
  ATOMNAS_REQUIRE(!dw || dw_ws, "dwconv_bwd: the weight gradient needs the partial workspace dw_ws [part_rows][C][k*k]");
  hipStream_t st = (hipStream_t)stream;
  {
    int rc = dwconv_mm_bwd(g, g_ss, yraw, yraw_ss, c1, c2, c3, x, x_ss, in_scale, in_shift, in_relu, w, ldw, h, h_ss, dw, stats, stat_ld,
                           part_rows, dw_ws, N, H, W, C, k, stride, dtype, st);
    if (rc >= 0) return rc;
    rc = dwconv_cw_bwd(g, g_ss, yraw, yraw_ss, c1, c2, c3, x, x_ss, in_scale, in_shift, in_relu, w, ldw, h, h_ss, dw, stats, stat_ld,
                       part_rows, dw_ws, N, H, W, C, k, stride, dtype, st);
    if (rc >= 0) return rc;
  }
  DW_DISPATCH(launch_bwd, g, ldg, g_ss, yraw, ldyr, yraw_ss, c1, c2, c3, x, ldx, x_ss, in_scale, in_shift, in_relu, w, ldw, h, ldh,
              h_ss, dw, stats, stat_ld, part_rows, dw_ws, N, H, W, C, st);
  return 1;
}

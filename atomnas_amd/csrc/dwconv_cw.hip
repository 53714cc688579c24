// Depthwise k x k convolution, stride 1, slab-major activations: the "one channel pair per wave" kernels (gfx950).
//
// Same operation and the same C-ABI entry points as dwconv.hip (models/mobilenet_base.py:330-336: the depthwise ConvBNReLU of
// the atomic block, forward and backward, with the surrounding BatchNorm / activation passes fused into the load side); this
// file holds the instances the hidden tensors of the expanding blocks take: stride 1, slab-major [C/16][M][16] tensors, image
// width a multiple of 7.  dwconv.hip keeps every other case (stride 2, plain layout, ragged widths).
//
// Why a second structure (round-2 profile of the tile kernels: the LDS pipe as busy as the FMA pipes, 2 waves per SIMD):
//   * a wave owns ONE channel pair; its 64 lanes are 64 strips of 7 pixels (image rows x strips of a row).  The k*k taps of
//     the pair are then wave-uniform and live in scalar registers (s_load), not in LDS: per tap row a lane reads only its
//     7 + k - 1 operand pairs (13 ds_read_b64 for k = 7, where the tile kernels read 20 of which the compiler merged pairs into
//     half-rate ds_read2_b64) for 98 packed FMAs, and no vector register holds a weight.
//   * the 8 waves of a 512-thread workgroup are the 8 channel pairs of one 16-channel slab; a tile is TH rows of the FULL image
//     width (or several whole images of a small map), so there is no horizontal halo, and the operand window is a ring over
//     rows, so there is no vertical halo either: every activation byte is read from HBM exactly once.
//   * LDS operand planes are [channel pair][row][column] with a row pitch chosen per strips-per-row so that the 32 lanes of an
//     LDS group start on 32 different 8-byte bank pairs: all operand reads are conflict-free ds_read_b64.
//   * the raw input pixels (backward) / the output tile (forward) pass through LDS as planes of packed channel pairs, so HBM is
//     touched with 16-byte accesses only; the backward result overwrites the raw input pixel in place (same lane, same slot).
//
// Numerics: identical arithmetic per element to dwconv.hip (fp32 accumulation, one rounding to the storage type); per-channel
// sums are per-lane partials added in a fixed order (lanes by butterfly, workers by reduce_parts / the BatchNorm finalize), so
// results are bit-reproducible.  No atomics.
#include "common.h"
#include <cstdlib>

namespace atomnas {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct CwGeom {
  int N, H, W, C;
  int TH, NI, tiles_y, ns;   // tile: NI images x TH rows x W columns; ns = W / 7 strips per row
  int LH, LWp, plane;        // operand window: rows per image, row pitch, elements (f32x2) per channel-pair plane
  int TPIX, TPIXp;           // pixels per tile; pitch of the pixel planes
  int nworkers, nslabs, ntiles;
  int ring;                  // 1: several tiles per image, the window rows are a ring
};

// storage-type plumbing: `piece` = 8 channels of one pixel (16-byte global accesses), `pair` = one channel pair of one pixel
template <typename T> struct Cw;
template <> struct Cw<bf16_t> {
  typedef unsigned pair_t;
  struct piece_t { u32x4 v; };
  static __device__ __forceinline__ void zero(piece_t& p) { p.v = u32x4{0u, 0u, 0u, 0u}; }
  static __device__ __forceinline__ void load(piece_t& p, const bf16_t* s) { p.v = *reinterpret_cast<const u32x4*>(s); }
  static __device__ __forceinline__ void store(const piece_t& p, bf16_t* d) { *reinterpret_cast<u32x4*>(d) = p.v; }
  static __device__ __forceinline__ pair_t pair(const piece_t& p, int q) { return p.v[q]; }
  static __device__ __forceinline__ void set_pair(piece_t& p, int q, pair_t v) { p.v[q] = v; }
  static __device__ __forceinline__ float lo(pair_t v) { return __uint_as_float(v << 16); }
  static __device__ __forceinline__ float hi(pair_t v) { return __uint_as_float(v & 0xffff0000u); }
  static __device__ __forceinline__ pair_t pack(float a, float b) {
    bf16x2 t;
    t[0] = (bf16_t)a; t[1] = (bf16_t)b;   // RNE
    return __builtin_bit_cast(unsigned, t);
  }
  static __device__ __forceinline__ pair_t zero_pair() { return 0u; }
};
template <> struct Cw<float> {
  typedef f32x2 pair_t;
  struct piece_t { f32x4 a, b; };
  static __device__ __forceinline__ void zero(piece_t& p) { p.a = f32x4{0.f, 0.f, 0.f, 0.f}; p.b = p.a; }
  static __device__ __forceinline__ void load(piece_t& p, const float* s) {
    p.a = *reinterpret_cast<const f32x4*>(s); p.b = *reinterpret_cast<const f32x4*>(s + 4);
  }
  static __device__ __forceinline__ void store(const piece_t& p, float* d) {
    *reinterpret_cast<f32x4*>(d) = p.a; *reinterpret_cast<f32x4*>(d + 4) = p.b;
  }
  static __device__ __forceinline__ pair_t pair(const piece_t& p, int q) {
    return q < 2 ? f32x2{p.a[2 * q], p.a[2 * q + 1]} : f32x2{p.b[2 * q - 4], p.b[2 * q - 3]};
  }
  static __device__ __forceinline__ void set_pair(piece_t& p, int q, pair_t v) {
    if (q < 2) { p.a[2 * q] = v[0]; p.a[2 * q + 1] = v[1]; } else { p.b[2 * q - 4] = v[0]; p.b[2 * q - 3] = v[1]; }
  }
  static __device__ __forceinline__ float lo(pair_t v) { return v[0]; }
  static __device__ __forceinline__ float hi(pair_t v) { return v[1]; }
  static __device__ __forceinline__ pair_t pack(float a, float b) { return f32x2{a, b}; }
  static __device__ __forceinline__ pair_t zero_pair() { return f32x2{0.f, 0.f}; }
};

__device__ __forceinline__ float cw_act(float a, int in_relu, int AM) {
  if (AM == ACT_RELU6) return fminf(fmaxf(a, 0.f), 6.f);
  if (AM == ACT_SWISH) return swish_f(a);
  return in_relu ? fmaxf(a, 0.f) : a;
}
__device__ __forceinline__ float cw_act_bwd(float c, float a, int in_relu, int AM) {
  if (AM == ACT_RELU6) return (a > 0.f && a < 6.f) ? c : 0.f;
  if (AM == ACT_SWISH) return c * swish_grad(a);
  return (in_relu && !(a > 0.f)) ? 0.f : c;
}

// value of lane `l` (compile-time constant) as a wave-uniform scalar
__device__ __forceinline__ float cw_bcast(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ unsigned cw_lds_addr(const void* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}
// One operand row of a lane: NR consecutive channel pairs (8 bytes each) from LDS byte address `addr`.  Inline asm on purpose:
// hipcc's load/store optimizer merges neighbouring ds_read_b64 into ds_read2_b64, which moves the same bytes in twice the LDS
// cycles (MI355X_MICROARCH.md, LDS table).  The reads are invisible to the compiler's wait counters: the wait and the scheduling
// fence below make every result valid before its first use (cdna_hip_programming.md 5.7 form iii).
template <int NR>
__device__ __forceinline__ void cw_read_row(f32x2 (&v)[NR], unsigned addr) {
#pragma unroll
  for (int i = 0; i < NR; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "i"(i * 8));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Tile-independent decode of the two staging slots of a thread: slot i of thread tid is 16-byte piece (tid + i * 512) of the tile,
// pieces run over (image, row, column, channel group) with the channel group fastest (512 is even: cg = tid & 1 for both slots).
struct CwSlots {
  int pp[2];     // pixel index inside the tile (im, row, col) -> also the index into the pixel planes; -1: no such piece
  int rr[2];     // row inside the tile
  int dyo[2];    // window element offset without the row term: im * LH * LWp + col + P
  int goff[2];   // element offset inside the slab relative to the tile's first pixel: ((im * H + rr) * W + col) * 16 + cg * 8
  int im[2];
};
template <int P>
__device__ __forceinline__ void cw_decode(CwSlots& s, const CwGeom& g, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = (tid + i * 512) >> 1;
    const bool ok = pp < g.TPIX;
    const int col = pp % g.W, t2 = pp / g.W;
    const int rr = t2 % g.TH, im = t2 / g.TH;
    s.pp[i] = ok ? pp : -1;
    s.rr[i] = rr;
    s.im[i] = im;
    s.dyo[i] = im * g.LH * g.LWp + col + P;
    s.goff[i] = ((im * g.H + rr) * g.W + col) * 16 + (tid & 1) * 8;
  }
}

// ---------------------------------------------------------------------------------------------------------------- backward
//   dYraw = c1*g + c2*yraw + c3 (BN-backward of the BN behind the conv, on load; yraw == NULL: dYraw = g)
//   h = dwconv^T(dYraw) * act'(x*in_scale+in_shift),  dW += corr(act(x*in_scale+in_shift), dYraw),  stats: sum h, sum h*x
template <typename T, int K, int AM, int WPS>
__global__ __launch_bounds__(512, WPS) void k_dwb_cw(const T* __restrict__ gup, long gss, const T* __restrict__ yraw, long yrss,
                                                     const float* __restrict__ c1, const float* __restrict__ c2p,
                                                     const float* __restrict__ c3, const T* __restrict__ x, long xss,
                                                     const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                     int in_relu, const float* __restrict__ w, int ldw, T* __restrict__ h, long hss,
                                                     float* __restrict__ dwp, float* __restrict__ stats, int stat_ld, int stat_rows,
                                                     CwGeom g) {
  typedef Cw<T> X;
  typedef typename X::pair_t pair_t;
  typedef typename X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, KK = K * K, SW = 7, DWN = SW + K - 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* s_dy = reinterpret_cast<f32x2*>(smem);                    // [8 pairs][plane]: dYraw window, fp32
  pair_t* s_x = reinterpret_cast<pair_t*>(s_dy + 8 * g.plane);      // [8 pairs][TPIXp]: raw input pixels, replaced by h in place
  float* s_cf = reinterpret_cast<float*>(s_x + 8 * g.TPIXp);        // [3][16] BN-backward coefficients of the slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);           // channel pair of this wave (wave-uniform)
  const int slab = blockIdx.x % g.nslabs, worker = blockIdx.x / g.nslabs;
  const int c_base = slab * 16;
  const int ch = c_base + 2 * wv;
  const int cpad = (g.C + 7) & ~7;
  const int cg = tid & 1;
  const bool cg_ok = c_base + cg * 8 < cpad;

  for (int i = tid; i < 8 * g.plane; i += 512) s_dy[i] = f32x2{0.f, 0.f};   // halo columns / rows outside the image stay zero
  if (tid < 48) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    const float* src = (v == 0) ? c1 : (v == 1 ? c2p : c3);
    s_cf[tid] = (c1 && src && (v == 0 || yraw) && c < cpad) ? src[c] : (v == 0 ? 1.f : 0.f);
  }

  // wave-uniform per-channel scalars
  float sc0 = 1.f, sc1 = 1.f, sh0 = 0.f, sh1 = 0.f;
  if (in_scale && ch < cpad) { sc0 = in_scale[ch]; sc1 = in_scale[ch + 1]; sh0 = in_shift[ch]; sh1 = in_shift[ch + 1]; }
  const bool ch0_ok = ch < g.C, ch1_ok = ch + 1 < g.C;
  // tap table of the pair: lane t (< k*k) holds tap t of both channels; the tap loop broadcasts them with v_readlane (constant lane)
  // into scalar registers -- two vector registers and one instruction per scalar instead of k*k LDS reads or k*k loads per row
  float wl0 = 0.f, wl1 = 0.f;
  if (lane < KK) {
    if (ch0_ok) wl0 = w[(long)lane * ldw + ch];
    if (ch1_ok) wl1 = w[(long)lane * ldw + ch + 1];
  }

  CwSlots sl;
  cw_decode<P>(sl, g, tid);

  // work item of this lane: (image, row, strip) -- tile-independent
  const int ipi = g.TH * g.ns;
  const int it_im = lane / ipi, it_rem = lane % ipi;
  const int it_r = it_rem / g.ns, it_j = it_rem % g.ns;
  const bool it_ok = lane < g.NI * ipi;
  const int pix0 = (it_im * g.TH + it_r) * g.W + SW * it_j;
  const unsigned dy_addr0 = cw_lds_addr(s_dy + wv * g.plane + it_im * g.LH * g.LWp + SW * it_j);   // + slot * LWp * 8

  f32x2 dwa[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) dwa[t] = f32x2{0.f, 0.f};
  float s0a = 0.f, s0b = 0.f, s1a = 0.f, s1b = 0.f;

  piece_t pfg[2], pfy[2], pfx[2];
  unsigned pfmask = 0, pxmask = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) { X::zero(pfg[i]); X::zero(pfy[i]); X::zero(pfx[i]); }

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_g = (long)slab * gss, slab_y = (long)slab * yrss, slab_x = (long)slab * xss, slab_h = (long)slab * hss;

  // issue the HBM loads of a tile (n0 = first image, hi0 = first row): dY rows [ho_s, ho_s + TH), input rows [hi0, hi0 + TH)
  auto issue = [&](int n0, int hi0) {
    const int ho_s = g.ring ? hi0 + P : 0;
    const long pg = ((long)n0 * g.H + ho_s) * g.W * 16, px = ((long)n0 * g.H + hi0) * g.W * 16;
    pfmask = 0; pxmask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N) {
        if (ho_s + sl.rr[i] < g.H) {
          X::load(pfg[i], gup + slab_g + pg + sl.goff[i]);
          if (yraw) X::load(pfy[i], yraw + slab_y + pg + sl.goff[i]);
          pfmask |= 1u << i;
        }
        if (hi0 + sl.rr[i] < g.H) {
          X::load(pfx[i], x + slab_x + px + sl.goff[i]);
          pxmask |= 1u << i;
        }
      }
    }
  };
  // dYraw of one piece -> the four pair planes of this thread's channel group
  auto put_dy = [&](const piece_t& pg_, const piece_t& py_, bool ok, f32x2* d) {
    float q1[8], q2[8], q3[8];
    VecIO<float, 8>::load(s_cf + cg * 8, q1);
    VecIO<float, 8>::load(s_cf + 16 + cg * 8, q2);
    VecIO<float, 8>::load(s_cf + 32 + cg * 8, q3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const pair_t gq = X::pair(pg_, q), yq = X::pair(py_, q);
      const float a0 = q1[2 * q] * X::lo(gq) + (q2[2 * q] * X::lo(yq) + q3[2 * q]);
      const float a1 = q1[2 * q + 1] * X::hi(gq) + (q2[2 * q + 1] * X::hi(yq) + q3[2 * q + 1]);
      d[q * g.plane] = ok ? f32x2{a0, a1} : f32x2{0.f, 0.f};
    }
  };
  auto commit = [&](int base) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0) {
        int slot = (g.ring ? 2 * P : P) + sl.rr[i] + base;
        if (slot >= g.LH) slot -= g.LH;
        put_dy(pfg[i], pfy[i], (pfmask >> i) & 1u, s_dy + (cg * 4) * g.plane + sl.dyo[i] + slot * g.LWp);
        pair_t* dx_ = s_x + (cg * 4) * g.TPIXp + sl.pp[i];
        const bool okx = (pxmask >> i) & 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) dx_[q * g.TPIXp] = okx ? X::pair(pfx[i], q) : X::zero_pair();
      }
    }
  };
  // first tile of an image (or of this worker): the 2P window rows above the tile's own rows, loaded synchronously
  auto halo_sync = [&](int n0, int hi0) {
    const int npc = 2 * P * g.W * 2;
    for (int p = tid; p < npc; p += 512) {
      const int col = (p >> 1) % g.W, wr = (p >> 1) / g.W;
      const int ho = hi0 - P + wr;
      piece_t a, b;
      X::zero(a); X::zero(b);
      const bool ok = cg_ok && ho >= 0 && ho < g.H && n0 < g.N;
      if (ok) {
        const long off = (((long)n0 * g.H + ho) * g.W + col) * 16 + cg * 8;
        X::load(a, gup + slab_g + off);
        if (yraw) X::load(b, yraw + slab_y + off);
      }
      put_dy(a, b, ok, s_dy + (cg * 4) * g.plane + wr * g.LWp + col + P);
    }
  };
  auto store_h = [&](int n0, int hi0) {
    const long px = ((long)n0 * g.H + hi0) * g.W * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N && hi0 + sl.rr[i] < g.H) {
        piece_t v;
        const pair_t* sx_ = s_x + (cg * 4) * g.TPIXp + sl.pp[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) X::set_pair(v, q, sx_[q * g.TPIXp]);
        X::store(v, h + slab_h + px + sl.goff[i]);
      }
    }
  };

  int tile = t_beg;
  int nb = tile / g.tiles_y, ty = tile % g.tiles_y;
  if (tile < t_end) issue(nb * g.NI, ty * g.TH);
  int base = 0;
  int pn0 = -1, phi0 = 0;   // tile whose result waits in s_x
  for (; tile < t_end; ++tile) {
    const int n0 = nb * g.NI, hi0 = ty * g.TH;
    const bool fresh = g.ring && (tile == t_beg || ty == 0);
    if (fresh) base = 0;
    __syncthreads();   // (A) previous tile consumed, its h complete in s_x (first pass: also orders the LDS initialisation)
    if (pn0 >= 0) store_h(pn0, phi0);
    commit(base);
    if (fresh) halo_sync(n0, hi0);
    __syncthreads();   // (B) window and pixel planes complete
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.TH);

    // opaque per tile (in uniform control flow: every lane of the table stays defined): the 2 k^2 broadcasts stay in their tap rows
    // instead of being hoisted out of the tile loop, where they spill
    asm volatile("" : "+v"(wl0), "+v"(wl1));
    if (it_ok && n0 + it_im < g.N && hi0 + it_r < g.H && ch < cpad) {
      pair_t* xp = s_x + wv * g.TPIXp + pix0;
      pair_t xq[SW];
      f32x2 xa[SW], dx[SW];
#pragma unroll
      for (int t = 0; t < SW; ++t) xq[t] = xp[t];
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        xa[t] = f32x2{cw_act(X::lo(xq[t]) * sc0 + sh0, in_relu, AM), cw_act(X::hi(xq[t]) * sc1 + sh1, in_relu, AM)};
        dx[t] = f32x2{0.f, 0.f};
        asm volatile("" : "+v"(xa[t]));   // computed here, not sunk behind the tap rows (that keeps every operand row alive)
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        int slot = it_r + (K - 1 - ky) + base;
        if (slot >= g.LH) slot -= g.LH;
        f32x2 dy[DWN];
        cw_read_row<DWN>(dy, dy_addr0 + (unsigned)(slot * g.LWp) * 8u);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const f32x2 wv2 = f32x2{cw_bcast(wl0, ky * K + kx), cw_bcast(wl1, ky * K + kx)};
#pragma unroll
          for (int t = 0; t < SW; ++t) {
            dx[t] += dy[t + (K - 1 - kx)] * wv2;
            dwa[ky * K + kx] += xa[t] * dy[t + (K - 1 - kx)];
          }
          asm volatile("" : "+v"(dwa[ky * K + kx]));   // this row's FMAs are done before the next row's reads are issued
        }
#pragma unroll
        for (int t = 0; t < SW; ++t) asm volatile("" : "+v"(dx[t]));
        __builtin_amdgcn_sched_barrier(0);   // one operand row live at a time (the tap loop is unrolled for static dwa indices)
      }
      // epilogue: activation backward of the producer, rounding, statistics; h replaces x in its LDS slot
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        const float x0 = X::lo(xq[t]), x1 = X::hi(xq[t]);
        float v0 = cw_act_bwd(dx[t][0], x0 * sc0 + sh0, in_relu, AM);
        float v1 = cw_act_bwd(dx[t][1], x1 * sc1 + sh1, in_relu, AM);
        v0 = ch0_ok ? v0 : 0.f;
        v1 = ch1_ok ? v1 : 0.f;
        const pair_t o = X::pack(v0, v1);
        v0 = X::lo(o); v1 = X::hi(o);   // statistics of the stored (rounded) values
        s0a += v0; s0b += v1;
        s1a += v0 * x0; s1b += v1 * x1;
        xp[t] = o;
      }
    }
    pn0 = n0; phi0 = hi0;
    nb = nnb; ty = nty;
    if (g.ring) { base += g.TH; if (base >= g.LH) base -= g.LH; }
  }
  __syncthreads();
  if (pn0 >= 0) store_h(pn0, phi0);

  // all 64 lanes of a wave hold the same channel pair: butterfly sums (fixed order), lane 0 owns row `worker` of the partial buffers
#pragma unroll
  for (int t = 0; t < KK; ++t) { dwa[t][0] = wave_sum(dwa[t][0]); dwa[t][1] = wave_sum(dwa[t][1]); }
  s0a = wave_sum(s0a); s0b = wave_sum(s0b); s1a = wave_sum(s1a); s1b = wave_sum(s1b);
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = ch + e;
      if (c < g.C) {
        if (dwp) {
          float* d = dwp + ((long)worker * g.C + c) * KK;
#pragma unroll
          for (int t = 0; t < KK; ++t) d[t] = dwa[t][e];
        }
        if (stats) {
          const float v0 = e ? s0b : s0a, v1 = e ? s1b : s1a;
          float* r = stats + (long)worker * 2 * stat_ld;
          r[c] = v0;
          r[stat_ld + c] = v1;
          stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, c);
          stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, (long)stat_ld + c);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- forward
//   y = dwconv(act(x*in_scale+in_shift)),  stats: sum y, sum y^2 (of the stored values)
template <typename T, int K, int AM, int WPS>
__global__ __launch_bounds__(512, WPS) void k_dwf_cw(const T* __restrict__ x, long xss, const float* __restrict__ in_scale,
                                                     const float* __restrict__ in_shift, int in_relu, const float* __restrict__ w,
                                                     int ldw, T* __restrict__ y, long yss, float* __restrict__ stats, int stat_ld,
                                                     int stat_rows, CwGeom g) {
  typedef Cw<T> X;
  typedef typename X::pair_t pair_t;
  typedef typename X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, SW = 7, IWN = SW + K - 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* s_in = reinterpret_cast<f32x2*>(smem);                     // [8 pairs][plane]: activated input window, fp32
  pair_t* s_y = reinterpret_cast<pair_t*>(s_in + 8 * g.plane);       // [8 pairs][TPIXp]: the tile's output
  float* s_cf = reinterpret_cast<float*>(s_y + 8 * g.TPIXp);         // [2][16] scale / shift of the slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.x % g.nslabs, worker = blockIdx.x / g.nslabs;
  const int c_base = slab * 16;
  const int ch = c_base + 2 * wv;
  const int cpad = (g.C + 7) & ~7;
  const int cg = tid & 1;
  const bool cg_ok = c_base + cg * 8 < cpad;

  for (int i = tid; i < 8 * g.plane; i += 512) s_in[i] = f32x2{0.f, 0.f};
  if (tid < 32) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    s_cf[tid] = (in_scale && c < cpad) ? (v == 0 ? in_scale[c] : in_shift[c]) : (v == 0 ? 1.f : 0.f);
  }
  const bool ch0_ok = ch < g.C, ch1_ok = ch + 1 < g.C;
  constexpr int KK = K * K;
  float wl0 = 0.f, wl1 = 0.f;   // tap table of the pair, see k_dwb_cw
  if (lane < KK) {
    if (ch0_ok) wl0 = w[(long)lane * ldw + ch];
    if (ch1_ok) wl1 = w[(long)lane * ldw + ch + 1];
  }

  CwSlots sl;
  cw_decode<P>(sl, g, tid);
  const int ipi = g.TH * g.ns;
  const int it_im = lane / ipi, it_rem = lane % ipi;
  const int it_r = it_rem / g.ns, it_j = it_rem % g.ns;
  const bool it_ok = lane < g.NI * ipi;
  const int pix0 = (it_im * g.TH + it_r) * g.W + SW * it_j;
  const unsigned in_addr0 = cw_lds_addr(s_in + wv * g.plane + it_im * g.LH * g.LWp + SW * it_j);

  float sa = 0.f, sb = 0.f, qa = 0.f, qb = 0.f;
  piece_t pfx[2];
  unsigned pxmask = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) X::zero(pfx[i]);

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_x = (long)slab * xss, slab_y = (long)slab * yss;

  auto issue = [&](int n0, int ho0) {
    const int hi_s = g.ring ? ho0 + P : 0;
    const long px = ((long)n0 * g.H + hi_s) * g.W * 16;
    pxmask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N && hi_s + sl.rr[i] < g.H) {
        X::load(pfx[i], x + slab_x + px + sl.goff[i]);
        pxmask |= 1u << i;
      }
    }
  };
  auto put_in = [&](const piece_t& p, bool ok, f32x2* d) {
    float q1[8], q2[8];
    VecIO<float, 8>::load(s_cf + cg * 8, q1);
    VecIO<float, 8>::load(s_cf + 16 + cg * 8, q2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const pair_t xq = X::pair(p, q);
      float a0 = X::lo(xq) * q1[2 * q] + q2[2 * q], a1 = X::hi(xq) * q1[2 * q + 1] + q2[2 * q + 1];
      a0 = cw_act(a0, in_relu, AM); a1 = cw_act(a1, in_relu, AM);
      d[q * g.plane] = ok ? f32x2{a0, a1} : f32x2{0.f, 0.f};
    }
  };
  auto commit = [&](int base) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0) {
        int slot = (g.ring ? 2 * P : P) + sl.rr[i] + base;
        if (slot >= g.LH) slot -= g.LH;
        put_in(pfx[i], (pxmask >> i) & 1u, s_in + (cg * 4) * g.plane + sl.dyo[i] + slot * g.LWp);
      }
    }
  };
  auto halo_sync = [&](int n0, int ho0) {
    const int npc = 2 * P * g.W * 2;
    for (int p = tid; p < npc; p += 512) {
      const int col = (p >> 1) % g.W, wr = (p >> 1) / g.W;
      const int hi = ho0 - P + wr;
      piece_t a;
      X::zero(a);
      const bool ok = cg_ok && hi >= 0 && hi < g.H && n0 < g.N;
      if (ok) X::load(a, x + slab_x + (((long)n0 * g.H + hi) * g.W + col) * 16 + cg * 8);
      put_in(a, ok, s_in + (cg * 4) * g.plane + wr * g.LWp + col + P);
    }
  };
  auto store_y = [&](int n0, int ho0) {
    const long py = ((long)n0 * g.H + ho0) * g.W * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N && ho0 + sl.rr[i] < g.H) {
        piece_t v;
        const pair_t* sy_ = s_y + (cg * 4) * g.TPIXp + sl.pp[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) X::set_pair(v, q, sy_[q * g.TPIXp]);
        X::store(v, y + slab_y + py + sl.goff[i]);
      }
    }
  };

  int tile = t_beg;
  int nb = tile / g.tiles_y, ty = tile % g.tiles_y;
  if (tile < t_end) issue(nb * g.NI, ty * g.TH);
  int base = 0;
  int pn0 = -1, pho0 = 0;
  for (; tile < t_end; ++tile) {
    const int n0 = nb * g.NI, ho0 = ty * g.TH;
    const bool fresh = g.ring && (tile == t_beg || ty == 0);
    if (fresh) base = 0;
    __syncthreads();   // (A) previous tile consumed, its output complete in s_y
    if (pn0 >= 0) store_y(pn0, pho0);
    commit(base);
    if (fresh) halo_sync(n0, ho0);
    __syncthreads();   // (B)
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.TH);

    asm volatile("" : "+v"(wl0), "+v"(wl1));   // see k_dwb_cw
    if (it_ok && n0 + it_im < g.N && ho0 + it_r < g.H && ch < cpad) {
      f32x2 acc[SW];
#pragma unroll
      for (int t = 0; t < SW; ++t) acc[t] = f32x2{0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        int slot = it_r + ky + base;
        if (slot >= g.LH) slot -= g.LH;
        f32x2 in[IWN];
        cw_read_row<IWN>(in, in_addr0 + (unsigned)(slot * g.LWp) * 8u);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const f32x2 wv2 = f32x2{cw_bcast(wl0, ky * K + kx), cw_bcast(wl1, ky * K + kx)};
#pragma unroll
          for (int t = 0; t < SW; ++t) acc[t] += in[t + kx] * wv2;
        }
#pragma unroll
        for (int t = 0; t < SW; ++t) asm volatile("" : "+v"(acc[t]));   // this row's FMAs are done before the next row's reads
        __builtin_amdgcn_sched_barrier(0);   // one operand row live at a time
      }
      pair_t* yp = s_y + wv * g.TPIXp + pix0;
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        const pair_t o = X::pack(ch0_ok ? acc[t][0] : 0.f, ch1_ok ? acc[t][1] : 0.f);
        const float v0 = X::lo(o), v1 = X::hi(o);
        sa += v0; sb += v1; qa += v0 * v0; qb += v1 * v1;
        yp[t] = o;
      }
    }
    pn0 = n0; pho0 = ho0;
    nb = nnb; ty = nty;
    if (g.ring) { base += g.TH; if (base >= g.LH) base -= g.LH; }
  }
  __syncthreads();
  if (pn0 >= 0) store_y(pn0, pho0);

  if (stats) {
    sa = wave_sum(sa); sb = wave_sum(sb); qa = wave_sum(qa); qb = wave_sum(qb);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = ch + e;
        if (c < g.C) {
          float* r = stats + (long)worker * 2 * stat_ld;
          r[c] = e ? sb : sa;
          r[stat_ld + c] = e ? qb : qa;
          stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, c);
          stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, (long)stat_ld + c);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- host side
static bool cw_geometry(CwGeom& g, int N, int H, int W, int C, int K) {
  if (W % 7 != 0 || W < 7) return false;
  g.N = N; g.H = H; g.W = W; g.C = C;
  g.ns = W / 7;
  if (g.ns > 16) return false;
  if (H * g.ns <= 64) {   // whole images
    g.TH = H; g.tiles_y = 1; g.NI = 64 / (H * g.ns); g.ring = 0;
    if (g.NI > N) g.NI = N;
  } else {
    const int cap = 64 / g.ns;
    const int nty = (H + cap - 1) / cap;
    g.TH = (H + nty - 1) / nty;
    g.tiles_y = (H + g.TH - 1) / g.TH;
    g.NI = 1; g.ring = 1;
  }
  g.LH = g.TH + K - 1;
  // row pitch: the 32 lanes of an LDS group are (rows x strips); their first elements r * LWp + 7 * j must differ mod 32 (8-byte
  // bank pairs): LWp = ns (mod 2 ns) for ns a power of two does it (7 is invertible mod 32), an odd pitch otherwise
  const int lw = W + K - 1;
  const bool pow2 = (g.ns & (g.ns - 1)) == 0;
  int lwp = lw;
  if (pow2) { while (lwp % (2 * g.ns) != g.ns) ++lwp; } else if (lwp % 2 == 0) ++lwp;
  g.LWp = lwp;
  int plane = g.NI * g.LH * g.LWp;
  while (plane % 4 != 2) ++plane;     // staging writes of the two channel groups land in different bank halves
  g.plane = plane;
  g.TPIX = g.NI * g.TH * W;
  int tp = g.TPIX;
  while (tp % 8 != 4) ++tp;
  g.TPIXp = tp;
  g.ntiles = ((N + g.NI - 1) / g.NI) * g.tiles_y;
  g.nslabs = (C + 15) / 16;
  return true;
}

static void cw_workers(CwGeom& g, int per_cu, int max_rows) {
  if (per_cu < 1) per_cu = 1;
  long want = ((long)num_cus() * per_cu) / g.nslabs;
  static const long max_env = getenv("ATOMNAS_DW_MAX_WORKERS") ? atol(getenv("ATOMNAS_DW_MAX_WORKERS")) : 0;   // tests: long tile walks
  if (max_env > 0 && want > max_env) want = max_env;
  if (max_rows > 0 && want > max_rows) want = max_rows;   // every worker owns one partial row
  if (want > g.ntiles) want = g.ntiles;
  if (want < 1) want = 1;
  g.nworkers = (int)want;
}

static int cw_mode() {
  static const int m = getenv("ATOMNAS_DW_CW") ? atoi(getenv("ATOMNAS_DW_CW")) : 3;   // bit 0: backward, bit 1: forward
  return m;
}

template <typename T, int K>
static int cw_launch_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                         const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                         float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  if (!cw_geometry(g, N, H, W, C, K)) return -1;
  typedef typename Cw<T>::pair_t pair_t;
  const size_t lds = (size_t)8 * g.plane * sizeof(f32x2) + (size_t)8 * g.TPIXp * sizeof(pair_t) + 48 * sizeof(float);
  if (lds > 160 * 1024) return -1;
  static const int wps_env = getenv("ATOMNAS_DW_CW_WPS") ? atoi(getenv("ATOMNAS_DW_CW_WPS")) : 0;
  // two workgroups per CU (128 registers) only where the 18 + ... accumulators of k = 3 fit and the LDS allows it
  const bool two = wps_env ? wps_env == 4 : (K == 3 && 2 * lds + 2048 <= 160 * 1024);
#define CW_BWD(AMV, WPSV)                                                                                                   \
  {                                                                                                                         \
    auto kern = k_dwb_cw<T, K, AMV, WPSV>;                                                                                  \
    cw_workers(g, resident_per_cu(kern, 512, lds), (stats || dw) ? part_rows : 0);                                          \
    hipLaunchKernelGGL(kern, dim3(g.nworkers * g.nslabs), dim3(512), lds, st, (const T*)gup, gss, (const T*)yraw, yrss, c1, c2, \
                       c3, (const T*)x, xss, sc, sh, relu, w, ldw, (T*)h, hss, dw ? dw_ws : nullptr, stats, stat_ld, part_rows, g); \
  }
  if (two) {
    if (relu == ACT_RELU6) CW_BWD(ACT_RELU6, 4) else if (relu == ACT_SWISH) CW_BWD(ACT_SWISH, 4) else CW_BWD(0, 4)
  } else {
    if (relu == ACT_RELU6) CW_BWD(ACT_RELU6, 2) else if (relu == ACT_SWISH) CW_BWD(ACT_SWISH, 2) else CW_BWD(0, 2)
  }
#undef CW_BWD
  if (int rc = check_launch("dwconv_bwd(cw)")) return rc;
  if (dw) return reduce_parts(dw_ws, (long)C * K * K, g.nworkers, (long)C * K * K, dw, C * K * K, 0, 1, st);
  return 0;
}

template <typename T, int K>
static int cw_launch_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                         float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  if (!cw_geometry(g, N, H, W, C, K)) return -1;
  typedef typename Cw<T>::pair_t pair_t;
  const size_t lds = (size_t)8 * g.plane * sizeof(f32x2) + (size_t)8 * g.TPIXp * sizeof(pair_t) + 32 * sizeof(float);
  if (lds > 160 * 1024) return -1;
  static const int wps_env = getenv("ATOMNAS_DW_CW_WPS_FWD") ? atoi(getenv("ATOMNAS_DW_CW_WPS_FWD")) : 0;
  const bool two = wps_env ? wps_env == 4 : (2 * lds + 2048 <= 160 * 1024);
#define CW_FWD(AMV, WPSV)                                                                                                   \
  {                                                                                                                         \
    auto kern = k_dwf_cw<T, K, AMV, WPSV>;                                                                                  \
    cw_workers(g, resident_per_cu(kern, 512, lds), stats ? stat_rows : 0);                                                  \
    hipLaunchKernelGGL(kern, dim3(g.nworkers * g.nslabs), dim3(512), lds, st, (const T*)x, xss, sc, sh, relu, w, ldw, (T*)y, yss, \
                       stats, stat_ld, stat_rows, g);                                                                       \
  }
  if (two) {
    if (relu == ACT_RELU6) CW_FWD(ACT_RELU6, 4) else if (relu == ACT_SWISH) CW_FWD(ACT_SWISH, 4) else CW_FWD(0, 4)
  } else {
    if (relu == ACT_RELU6) CW_FWD(ACT_RELU6, 2) else if (relu == ACT_SWISH) CW_FWD(ACT_SWISH, 2) else CW_FWD(0, 2)
  }
#undef CW_FWD
  return check_launch("dwconv_fwd(cw)");
}

// -1: not one of this file's cases (the caller continues with the tile kernels of dwconv.hip); otherwise the launch status
int dwconv_cw_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                  const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                  float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k, int dtype,
                  hipStream_t st) {
  if (!(cw_mode() & 1) || gss == 0 || xss == 0 || hss == 0 || (yraw && yrss == 0)) return -1;
#define CW_B(TT, KV) return cw_launch_bwd<TT, KV>(gup, gss, yraw, yrss, c1, c2, c3, x, xss, sc, sh, relu, w, ldw, h, hss, dw, stats, stat_ld, part_rows, dw_ws, N, H, W, C, st)
  if (dtype == DT_F32) {
    if (k == 3) CW_B(float, 3); if (k == 5) CW_B(float, 5); if (k == 7) CW_B(float, 7);
  } else {
    if (k == 3) CW_B(bf16_t, 3); if (k == 5) CW_B(bf16_t, 5); if (k == 7) CW_B(bf16_t, 7);
  }
#undef CW_B
  return -1;
}

int dwconv_cw_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                  float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k, int dtype, hipStream_t st) {
  if (!(cw_mode() & 2) || xss == 0 || yss == 0) return -1;
#define CW_F(TT, KV) return cw_launch_fwd<TT, KV>(x, xss, sc, sh, relu, w, ldw, y, yss, stats, stat_ld, stat_rows, N, H, W, C, st)
  if (dtype == DT_F32) {
    if (k == 3) CW_F(float, 3); if (k == 5) CW_F(float, 5); if (k == 7) CW_F(float, 7);
  } else {
    if (k == 3) CW_F(bf16_t, 3); if (k == 5) CW_F(bf16_t, 5); if (k == 7) CW_F(bf16_t, 7);
  }
#undef CW_F
  return -1;
}

}  // namespace atomnas

// 1 when atomnas_dwconv_fwd (dir = 0) / atomnas_dwconv_bwd (dir = 1) take the channel-pair-per-wave kernels of this file for the
// shape (slab-major tensors, stride 1), 0 when they take the tile kernels of dwconv.hip.  Tests and launch-geometry tools only.
extern "C" int atomnas_dwconv_cw_supported(int N, int H, int W, int C, int k, int stride, int dtype, int dir) {
  using namespace atomnas;
  if (stride != 1 || !(k == 3 || k == 5 || k == 7) || !(cw_mode() & (dir ? 1 : 2))) return 0;
  CwGeom g;
  if (!cw_geometry(g, N, H, W, C, k)) return 0;
  const size_t pair = dtype == DT_F32 ? sizeof(f32x2) : sizeof(unsigned);
  const size_t lds = (size_t)8 * g.plane * sizeof(f32x2) + (size_t)8 * g.TPIXp * pair + 48 * sizeof(float);
  return lds <= 160 * 1024 ? 1 : 0;
}

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
//   * the waves of a workgroup are the channel pairs of one 16-channel slab (8 waves) or of its 8-channel half (4 waves, so that
//     two to four workgroups per CU overlap each other's load / commit / compute phases); a tile is TH rows of the FULL image
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
#include "dwconv_cw.h"

namespace atomnas {

#if CW_TIMING
__device__ unsigned long long g_cw_timing[8];
#endif

// ---------------------------------------------------------------------------------------------------------------- backward
//   dYraw = c1*g + c2*yraw + c3 (BN-backward of the BN behind the conv, on load; yraw == NULL: dYraw = g)
//   h = dwconv^T(dYraw) * act'(x*in_scale+in_shift),  dW += corr(act(x*in_scale+in_shift), dYraw),  stats: sum h, sum h*x
template <typename T, int K, int AM, int NW, int WPS, bool PF>
__global__ __launch_bounds__(NW * 64, WPS) void k_dwb_cw(const T* __restrict__ gup, long gss, const T* __restrict__ yraw, long yrss,
                                                     const float* __restrict__ c1, const float* __restrict__ c2p,
                                                     const float* __restrict__ c3, const T* __restrict__ x, long xss,
                                                     const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                     int in_relu, const float* __restrict__ w, int ldw, T* __restrict__ h, long hss,
                                                     float* __restrict__ dwp, float* __restrict__ stats, int stat_ld, int stat_rows,
                                                     CwGeom g) {
  typedef Cw<T> X;
  typedef typename X::pair_t pair_t;
  typedef typename X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, KK = K * K, SW = 7, DWN = SW + K - 1, NT = NW * 64, CGS = NW / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* s_dy = reinterpret_cast<f32x2*>(smem);                    // [NW pairs][plane]: dYraw window, fp32
  pair_t* s_x = reinterpret_cast<pair_t*>(s_dy + NW * g.plane);     // [NW pairs][TPIXp]: raw input pixels, replaced by h in place
  float* s_cf = reinterpret_cast<float*>(s_x + NW * g.TPIXp);       // [3][16] BN-backward coefficients of the slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);           // channel pair of this wave (wave-uniform)
  int slab, worker, half;
  if (!cw_block<NW>(g, slab, worker, half)) return;   // surplus block of the padded half-slab grid (whole workgroup, before any barrier)
  const int c_base = slab * 16;
  const int ch = c_base + 2 * (wv + 4 * half);
  const int cpad = (g.C + 7) & ~7;
  const int cgl = CGS == 2 ? (tid & 1) : 0;   // channel group inside the workgroup's planes
  const int cg = cgl + half;                   // channel group inside the slab
  const bool cg_ok = c_base + cg * 8 < cpad;

  for (int i = tid; i < NW * g.plane; i += NT) s_dy[i] = f32x2{0.f, 0.f};   // halo columns / rows outside the image stay zero
  if (tid < 48) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    const float* src = (v == 0) ? c1 : (v == 1 ? c2p : c3);
    s_cf[tid] = (c1 && src && (v == 0 || yraw) && c < cpad) ? src[c] : (v == 0 ? 1.f : 0.f);
  }

  // wave-uniform per-channel scalars
  float sc0 = 1.f, sc1 = 1.f, sh0 = 0.f, sh1 = 0.f;
  if (in_scale && ch < cpad) { sc0 = in_scale[ch]; sc1 = in_scale[ch + 1]; sh0 = in_shift[ch]; sh1 = in_shift[ch + 1]; }
  const bool ch0_ok = ch < g.C, ch1_ok = ch + 1 < g.C;
  // taps of the pair: w[t * ldw + ch], w[t * ldw + ch + 1] (ldw >= C rounded up to 8: the host side checks), read per tap row
  // as wave-uniform scalars.  Taps of channels beyond C are whatever the table holds there: their results are forced to zero.
  const float* wp = w + ch;
  unsigned ld4 = (unsigned)ldw * 4u;

  CwSlots sl;
  cw_decode<P, NT, CGS>(sl, g, tid, cg);

  // work item of this lane: (image, row, strip) -- tile-independent
  const int ipi = g.TH * g.ns;
  const int it_im = lane / ipi, it_rem = lane % ipi;
  const int it_r = it_rem / g.ns, it_j = it_rem % g.ns;
  const bool it_ok = lane < g.NI * ipi;
  const int pix0 = (it_im * g.TH + it_r) * g.W + SW * it_j;
  const unsigned dy_addr0 = cw_lds_addr(s_dy + wv * g.plane + it_im * g.RH * g.LWp + SW * it_j);   // + slot * LWp * 8

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
  // Branch-free: a piece that does not exist (image / row beyond the tensor, channel group beyond C) reads the first piece of the
  // slab instead and is masked at the commit.  (With the loads under per-lane branches hipcc cannot prove at the loop back-edge
  // that they were waited for, and puts an s_waitcnt vmcnt(0) in front of the next tile's loads: that wait also covers the h
  // stores issued just before -- 20 to 30 % of the kernel in the first measurements.)
  auto issue = [&](int n0, int hi0) {
    const int ho_s = g.ring ? hi0 + P : 0;
    const long pg = ((long)n0 * g.H + ho_s) * g.W * 16, px = ((long)n0 * g.H + hi0) * g.W * 16;
    pfmask = 0; pxmask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool in_n = sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N;
      const bool okg = in_n && ho_s + sl.rr[i] < g.H, okx = in_n && hi0 + sl.rr[i] < g.H;
      const long og = okg ? pg + sl.goff[i] : 0, ox = okx ? px + sl.goff[i] : 0;
      X::load(pfg[i], gup + slab_g + og);
      if (yraw) X::load(pfy[i], yraw + slab_y + og);
      X::load(pfx[i], x + slab_x + ox);
      pfmask |= okg ? 1u << i : 0u;
      pxmask |= okx ? 1u << i : 0u;
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
        put_dy(pfg[i], pfy[i], (pfmask >> i) & 1u, s_dy + (cgl * 4) * g.plane + sl.dyo[i] + slot * g.LWp);
        pair_t* dx_ = s_x + (cgl * 4) * g.TPIXp + sl.pp[i];
        const bool okx = (pxmask >> i) & 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) dx_[q * g.TPIXp] = okx ? X::pair(pfx[i], q) : X::zero_pair();
      }
    }
  };
  // first tile of an image (or of this worker): the 2P window rows above the tile's own rows, loaded synchronously
  auto halo_sync = [&](int n0, int hi0) {
    const int npc = 2 * P * g.W * CGS;
    for (int p = tid; p < npc; p += NT) {
      const int col = (p / CGS) % g.W, wr = (p / CGS) / g.W;
      const int ho = hi0 - P + wr;
      piece_t a, b;
      X::zero(a); X::zero(b);
      const bool ok = cg_ok && ho >= 0 && ho < g.H && n0 < g.N;
      if (ok) {
        const long off = (((long)n0 * g.H + ho) * g.W + col) * 16 + cg * 8;
        X::load(a, gup + slab_g + off);
        if (yraw) X::load(b, yraw + slab_y + off);
      }
      put_dy(a, b, ok, s_dy + (cgl * 4) * g.plane + wr * g.LWp + col + P);
    }
  };
  auto store_h = [&](int n0, int hi0) {
    const long px = ((long)n0 * g.H + hi0) * g.W * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N && hi0 + sl.rr[i] < g.H) {
        piece_t v;
        const pair_t* sx_ = s_x + (cgl * 4) * g.TPIXp + sl.pp[i];
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
#if CW_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  for (; tile < t_end; ++tile) {
    const int n0 = nb * g.NI, hi0 = ty * g.TH;
    const bool fresh = g.ring && (tile == t_beg || ty == 0);
    if (fresh) base = 0;
    CWMARK(6)
    // opaque per tile: the 64-bit global addresses of the slots are formed where they are used.  (Hoisted out of the tile loop they
    // are spilled, and every reload waits for vmcnt(0), i.e. for all loads and stores issued before it: the prefetch serialises.)
    asm volatile("" : "+v"(sl.goff[0]), "+v"(sl.goff[1]), "+v"(sl.pp[0]), "+v"(sl.pp[1]), "+v"(sl.dyo[0]), "+v"(sl.dyo[1]));
    __syncthreads();   // (A) previous tile consumed, its h complete in s_x (first pass: also orders the LDS initialisation)
    CWMARK(0)
    // every prefetched register is consumed here on every path: nothing is pending when the next tile's loads overwrite them
#pragma unroll
    for (int i = 0; i < 2; ++i) { X::touch(pfg[i]); X::touch(pfy[i]); X::touch(pfx[i]); }
    if (pn0 >= 0) store_h(pn0, phi0);
    commit(base);
    if (fresh) halo_sync(n0, hi0);
    CWMARK(1)
    __syncthreads();   // (B) window and pixel planes complete
    CWMARK(2)
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.TH);
    CWMARK(3)

    // opaque per tile: the k^2 tap offsets are formed in their tap rows instead of being hoisted out of the tile loop (they spill)
    asm volatile("" : "+s"(ld4));
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
      asm volatile("" ::: "memory");
      // tap rows.  PF (k = 7, where the registers allow two waves per SIMD anyway): the operands of row ky + 1 are in flight while
      // row ky is multiplied -- two operand buffers, static indices after unrolling.
      f32x2 dyb[PF ? 2 : 1][DWN], wb[PF ? 2 : 1][K];
      auto row_addr = [&](int ky) {
        int slot = it_r + (K - 1 - ky) + base;
        if (slot >= g.LH) slot -= g.LH;
        return dy_addr0 + (unsigned)(slot * g.LWp) * 8u;
      };
      if (PF) cw_row_issue<K, DWN>(wb[0], dyb[0], wp, 0u, ld4, row_addr(0));
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        constexpr int dummy = 0; (void)dummy;
        const int cur = PF ? (ky & 1) : 0;
        if (!PF) cw_row_issue<K, DWN>(wb[0], dyb[0], wp, (unsigned)(ky * K) * ld4, ld4, row_addr(ky));
        cw_row_wait();
        if (PF && ky + 1 < K) {
          cw_row_issue<K, DWN>(wb[cur ^ 1], dyb[cur ^ 1], wp, (unsigned)((ky + 1) * K) * ld4, ld4, row_addr(ky + 1));
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int t = 0; t < SW; ++t) {
            dx[t] += dyb[cur][t + (K - 1 - kx)] * wb[cur][kx];
            dwa[ky * K + kx] += xa[t] * dyb[cur][t + (K - 1 - kx)];
          }
          asm volatile("" : "+v"(dwa[ky * K + kx]));   // this row's FMAs are done before the next row's operands are touched
        }
#pragma unroll
        for (int t = 0; t < SW; ++t) asm volatile("" : "+v"(dx[t]));
        __builtin_amdgcn_sched_barrier(0);   // one / two operand rows live at a time (the tap loop is unrolled for static dwa indices)
      }
      // epilogue: activation backward of the producer, rounding, statistics; h replaces x in its LDS slot
#pragma unroll
      for (int t = 0; t < SW; ++t) xq[t] = xp[t];   // read again: the raw pixels are not kept in registers across the tap rows
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
    CWMARK(4)
    pn0 = n0; phi0 = hi0;
    nb = nnb; ty = nty;
    if (g.ring) { base += g.TH; if (base >= g.LH) base -= g.LH; }
  }
  __syncthreads();
  if (pn0 >= 0) store_h(pn0, phi0);
  CWMARK(5)

  // Weight-gradient flush.  All 64 lanes of a wave hold partial sums of the SAME 2 k^2 values (tap t of channel e = value e k^2 + t,
  // which is also its position in this wave's 2 k^2 consecutive floats of the partial row).  Cross-lane sums with 6 DPP steps per
  // value are ~1300 instructions of straight-line code that run once -- measured 30-45 us per launch, mostly instruction-cache
  // misses.  Instead: the wave transposes G = 14 values at a time through its own (now unused) operand plane -- lane l writes
  // row l of a [64][G + 1] matrix -- and lane q * G + v adds quarter q (16 lanes, in lane order) of value v; the four quarters are
  // added in order by lane v, which stores the total.  Fixed order, coalesced stores, a few hundred instructions.
  {
    constexpr int G = 14, NV = 2 * KK;
    float* red = reinterpret_cast<float*>(s_dy + wv * g.plane);   // 64 * (G + 1) + 4 * G floats <= 2 * plane (cw_geometry)
    float* red2 = red + 64 * (G + 1);
    float* drow = dwp ? dwp + ((long)worker * g.C + ch) * KK : nullptr;
    const int nvalid = ch1_ok ? NV : (ch0_ok ? KK : 0);
    const int rq = lane / G, rv = lane - rq * G;   // quarter and value of this lane in the column sums (lanes 0 .. 4G-1)
#pragma unroll
    for (int r0 = 0; r0 < NV; r0 += G) {
#pragma unroll
      for (int v = 0; v < G; ++v)
        if (r0 + v < NV) red[lane * (G + 1) + v] = (r0 + v < KK) ? dwa[(r0 + v) % KK][0] : dwa[(r0 + v) % KK][1];
      __builtin_amdgcn_wave_barrier();
      float part = 0.f;
      if (lane < 4 * G) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) part += red[(rq * 16 + i) * (G + 1) + rv];
        red2[lane] = part;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < G && drow && r0 + lane < nvalid) drow[r0 + lane] = ((red2[lane] + red2[G + lane]) + red2[2 * G + lane]) + red2[3 * G + lane];
      __builtin_amdgcn_wave_barrier();
    }
  }
  // BN-backward statistics of the pair: 4 values, DPP sums (fixed order), lane 63 owns row `worker` of the partial buffer
  s0a = cw_wave_sum63(s0a); s0b = cw_wave_sum63(s0b); s1a = cw_wave_sum63(s1a); s1b = cw_wave_sum63(s1b);
  if (lane == 63 && stats) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = ch + e;
      if (c < g.C) {
        const float v0 = e ? s0b : s0a, v1 = e ? s1b : s1a;
        float* r = stats + (long)worker * 2 * stat_ld;
        r[c] = v0;
        r[stat_ld + c] = v1;
        stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, c);
        stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, (long)stat_ld + c);
      }
    }
  }
#if CW_TIMING
  CWMARK(7)
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_cw_timing[i], tacc[i]);
  }
#endif
}


// ------------------------------------------------------------------------------------------------------- backward, stride 2
// Stride-2 depthwise backward in the same form.  The lanes live on the dY grid (Ho x Wo = H/2 x W/2): a lane owns 7 dY columns of
// one dY row, i.e. a 2 x 14 block of input pixels, and walks its two input rows one after the other.  Which taps meet which input
// pixel is a matter of row / column parity, and with whole 2 x 14 blocks per lane every parity test is a compile-time constant:
// no lane is masked (the tile kernels lose half of the lanes of every tap row of a stride-2 layer).  The window holds dYraw exactly as
// in k_dwb_cw (ring over dY rows, HL kept rows); the input tile is 4x the dY tile (2 THd x W pixels, contiguous in HBM because the
// tile spans the full width), staged through 7 register slots per thread that need no per-slot state.

template <typename T, int K, int AM, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void k_dwb_cw2(const T* __restrict__ gup, long gss, const T* __restrict__ yraw, long yrss,
                                                          const float* __restrict__ c1, const float* __restrict__ c2p,
                                                          const float* __restrict__ c3, const T* __restrict__ x, long xss,
                                                          const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                          int in_relu, const float* __restrict__ w, int ldw, T* __restrict__ h, long hss,
                                                          float* __restrict__ dwp, float* __restrict__ stats, int stat_ld,
                                                          int stat_rows, CwGeom g) {
  typedef Cw<T> X;
  typedef typename X::pair_t pair_t;
  typedef typename X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, KK = K * K, SW = 14, NT = NW * 64, CGS = NW / 4;
  constexpr int RELMIN = cw_fdiv(-P, 2), RELMAX = cw_fdiv(SW - 1 + P, 2), DWN = RELMAX - RELMIN + 1, CL = -RELMIN;
  constexpr int RT = P / 2;                              // dY rows above the first row pair of a tile that its taps reach
  constexpr int HL = RT + cw_fdiv(P - 1, 2) + 1;         // window rows shared with the tile above (LH = THd + HL)
  constexpr int XS = (448 * 4 * CGS + NT - 1) / NT;      // staging slots of the input tile (<= 64 lanes x 28 pixels)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* s_dy = reinterpret_cast<f32x2*>(smem);                    // [NW pairs][plane]
  pair_t* s_x = reinterpret_cast<pair_t*>(s_dy + NW * g.plane);     // [NW pairs][TPIXp]: raw input pixels, replaced by h in place
  float* s_cf = reinterpret_cast<float*>(s_x + NW * g.TPIXp);       // [3][16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int slab, worker, half;
  if (!cw_block<NW>(g, slab, worker, half)) return;
  const int c_base = slab * 16;
  const int ch = c_base + 2 * (wv + 4 * half);
  const int cpad = (g.C + 7) & ~7;
  const int cgl = CGS == 2 ? (tid & 1) : 0;
  const int cg = cgl + half;
  const bool cg_ok = c_base + cg * 8 < cpad;

  for (int i = tid; i < NW * g.plane; i += NT) s_dy[i] = f32x2{0.f, 0.f};
  if (tid < 48) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    const float* src = (v == 0) ? c1 : (v == 1 ? c2p : c3);
    s_cf[tid] = (c1 && src && (v == 0 || yraw) && c < cpad) ? src[c] : (v == 0 ? 1.f : 0.f);
  }
  float sc0 = 1.f, sc1 = 1.f, sh0 = 0.f, sh1 = 0.f;
  if (in_scale && ch < cpad) { sc0 = in_scale[ch]; sc1 = in_scale[ch + 1]; sh0 = in_shift[ch]; sh1 = in_shift[ch + 1]; }
  const bool ch0_ok = ch < g.C, ch1_ok = ch + 1 < g.C;
  const float* wp = w + ch;   // taps of the pair as wave-uniform scalars, see k_dwb_cw
  unsigned ld4 = (unsigned)ldw * 4u;

  // dY staging slots (two per thread): piece (tid + i NT) of the tile's dY rows, (image, row, column, channel group)
  int d_pp[2], d_rr[2], d_dyo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = (tid + i * NT) / CGS;
    d_pp[i] = pp < g.TPIXD ? pp : -1;
    const int col = pp % g.Wo, t2 = pp / g.Wo;
    d_rr[i] = t2 % g.THd;
    d_dyo[i] = (t2 / g.THd) * g.RH * g.LWp + col + CL;
  }
  int cgoff = cg * 8;   // channel-group offset inside a 16-channel pixel (opaque per tile, see k_dwb_cw)
  int tq = tid;         // thread id as the staging slots see it: opaque per tile, so that the 7 x 4 plane addresses and the 7 global
                        // offsets of the input slots are formed where they are used instead of living in registers across the tap rows

  // work item of this lane: (image, dY row, strip of 7 dY columns) = input rows 2 m, 2 m + 1, input columns 14 j .. 14 j + 13
  const int ipi = g.THd * g.ns;
  const int it_im = lane / ipi, it_rem = lane % ipi;
  const int it_m = it_rem / g.ns, it_j = it_rem % g.ns;
  const bool it_ok = lane < g.NI * ipi;
  const int pix00 = (it_im * g.TH + 2 * it_m) * g.W + SW * it_j;   // first pixel of the lane's upper input row inside the tile
  const unsigned dy_addr0 = cw_lds_addr(s_dy + wv * g.plane + it_im * g.RH * g.LWp + 7 * it_j);

  f32x2 dwa[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) dwa[t] = f32x2{0.f, 0.f};
  float s0a = 0.f, s0b = 0.f, s1a = 0.f, s1b = 0.f;

  piece_t pfg[2], pfy[2], pfx[XS];
  int pmaxd = 0, pmaxx = 0;   // valid dY pieces / input pixels of the prefetched tile (contiguous: everything below is valid)
#pragma unroll
  for (int i = 0; i < 2; ++i) { X::zero(pfg[i]); X::zero(pfy[i]); }
#pragma unroll
  for (int i = 0; i < XS; ++i) X::zero(pfx[i]);

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_g = (long)slab * gss, slab_y = (long)slab * yrss, slab_x = (long)slab * xss, slab_h = (long)slab * hss;

  // n0: first image, hd0: first dY row of the tile (input rows 2 hd0 ...).  Full-width tiles are contiguous in HBM: piece p of the
  // tile is at base + 16 p, and "valid" is p < a per-tile bound.
  auto issue = [&](int n0, int hd0) {
    const int ho_s = g.ring ? hd0 - RT + HL : 0;     // first dY row of the steady slots (window row HL, or RT for whole images)
    const long pg = ((long)n0 * g.Ho + ho_s) * g.Wo * 16, px = ((long)n0 * g.H + 2 * hd0) * g.W * 16;
    int vr = g.ring ? g.Ho - ho_s : g.Ho * (g.N - n0 < g.NI ? g.N - n0 : g.NI);   // valid dY rows from ho_s on
    if (g.ring && vr > g.THd) vr = g.THd;
    if (vr < 0) vr = 0;
    pmaxd = cg_ok ? vr * g.Wo : 0;
    int vx = g.ring ? g.H - 2 * hd0 : g.H * (g.N - n0 < g.NI ? g.N - n0 : g.NI);
    if (g.ring && vx > g.TH) vx = g.TH;
    pmaxx = cg_ok ? vx * g.W : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = d_pp[i] >= 0 && d_pp[i] < pmaxd;
      const long og = ok ? pg + (long)d_pp[i] * 16 + cgoff : 0;
      X::load(pfg[i], gup + slab_g + og);
      if (yraw) X::load(pfy[i], yraw + slab_y + og);
    }
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int pp = (tq + i * NT) / CGS;
      X::load(pfx[i], x + slab_x + (pp < pmaxx ? px + (long)pp * 16 + cgoff : 0));
    }
  };
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
      if (d_pp[i] >= 0) {
        int slot = (g.ring ? HL : RT) + d_rr[i] + base;
        if (slot >= g.LH) slot -= g.LH;
        put_dy(pfg[i], pfy[i], d_pp[i] < pmaxd, s_dy + (cgl * 4) * g.plane + d_dyo[i] + slot * g.LWp);
      }
    }
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int pp = (tq + i * NT) / CGS;
      if (pp < g.TPIX) {
        pair_t* dx_ = s_x + (cgl * 4) * g.TPIXp + pp;
        const bool okx = pp < pmaxx;
#pragma unroll
        for (int q = 0; q < 4; ++q) dx_[q * g.TPIXp] = okx ? X::pair(pfx[i], q) : X::zero_pair();
      }
    }
  };
  auto halo_sync = [&](int n0, int hd0) {   // first tile of an image / of this worker: the HL window rows above the tile's own rows
    const int npc = HL * g.Wo * CGS;
    for (int p = tid; p < npc; p += NT) {
      const int col = (p / CGS) % g.Wo, wr = (p / CGS) / g.Wo;
      const int ho = hd0 - RT + wr;
      piece_t a, b;
      X::zero(a); X::zero(b);
      const bool ok = cg_ok && ho >= 0 && ho < g.Ho && n0 < g.N;
      if (ok) {
        const long off = (((long)n0 * g.Ho + ho) * g.Wo + col) * 16 + cg * 8;
        X::load(a, gup + slab_g + off);
        if (yraw) X::load(b, yraw + slab_y + off);
      }
      put_dy(a, b, ok, s_dy + (cgl * 4) * g.plane + wr * g.LWp + col + CL);
    }
  };
  auto store_h = [&](int n0, int hd0, int pmax) {
    const long px = ((long)n0 * g.H + 2 * hd0) * g.W * 16;
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int pp = (tq + i * NT) / CGS;
      if (pp < pmax) {
        piece_t v;
        const pair_t* sx_ = s_x + (cgl * 4) * g.TPIXp + pp;
#pragma unroll
        for (int q = 0; q < 4; ++q) X::set_pair(v, q, sx_[q * g.TPIXp]);
        X::store(v, h + slab_h + px + (long)pp * 16 + cgoff);
      }
    }
  };

  int tile = t_beg;
  int nb = tile / g.tiles_y, ty = tile % g.tiles_y;
  if (tile < t_end) issue(nb * g.NI, ty * g.THd);
  int base = 0;
  int pn0 = -1, phd0 = 0, ppmax = 0;   // tile whose result waits in s_x, and its valid pixel count
  for (; tile < t_end; ++tile) {
    const int n0 = nb * g.NI, hd0 = ty * g.THd;
    const bool fresh = g.ring && (tile == t_beg || ty == 0);
    if (fresh) base = 0;
    asm volatile("" : "+v"(cgoff), "+v"(tq));
    __syncthreads();   // (A) previous tile consumed, its h complete in s_x
#pragma unroll
    for (int i = 0; i < 2; ++i) { X::touch(pfg[i]); X::touch(pfy[i]); }
#pragma unroll
    for (int i = 0; i < XS; ++i) X::touch(pfx[i]);
    if (pn0 >= 0) store_h(pn0, phd0, ppmax);
    commit(base);
    const int cur_pmaxx = pmaxx;
    if (fresh) halo_sync(n0, hd0);
    __syncthreads();   // (B) window and pixel planes complete
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    asm volatile("" : "+v"(tq));
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.THd);

    asm volatile("" : "+s"(ld4));
    if (it_ok && n0 + it_im < g.N && hd0 + it_m < g.Ho && ch < cpad) {
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {     // the lane's two input rows ...
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {   // ... each in two halves of 7 pixels: half the live operands (k = 7 spilled on whole strips)
          constexpr int HW7 = 7;
          // dY columns the half strip reaches: jj = (t + P - kx) / 2 - RELMIN over its pixels t and the taps of matching parity
          const int t0 = HW7 * sh;
          const int emin = t0 + P - (K - 1), emax = t0 + HW7 - 1 + P;
          const int jlo = cw_fdiv(emin + 1, 2) - RELMIN;                  // smallest even number >= emin, halved
          const int jhi = cw_fdiv(emax, 2) - RELMIN;                      // largest even number <= emax, halved
          constexpr int JN = (HW7 - 1 + K - 1) / 2 + 2;                   // enough for either half
          pair_t* xp = s_x + wv * g.TPIXp + pix00 + rp * g.W + t0;
          pair_t xq[HW7];
          f32x2 xa[HW7], dx[HW7];
#pragma unroll
          for (int t = 0; t < HW7; ++t) xq[t] = xp[t];
#pragma unroll
          for (int t = 0; t < HW7; ++t) {
            xa[t] = f32x2{cw_act(X::lo(xq[t]) * sc0 + sh0, in_relu, AM), cw_act(X::hi(xq[t]) * sc1 + sh1, in_relu, AM)};
            dx[t] = f32x2{0.f, 0.f};
            asm volatile("" : "+v"(xa[t]));
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            if ((rp + P - ky) & 1) continue;                  // this tap row meets rows of the other parity
            const int d = (rp + P - ky) / 2;                  // dY row relative to the lane's own: hd = m + d
            int slot = it_m + d + RT + base;
            if (slot >= g.LH) slot -= g.LH;
            f32x2 dy[JN], wr_[K];
            cw_row_issue<K, JN>(wr_, dy, wp, (unsigned)(ky * K) * ld4, ld4, dy_addr0 + (unsigned)(slot * g.LWp + jlo) * 8u);
            cw_row_wait();
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
#pragma unroll
              for (int t = 0; t < HW7; ++t) {
                if ((t0 + t + P - kx) & 1) continue;            // column parity
                const int jj = (t0 + t + P - kx) / 2 - RELMIN - jlo;
                dx[t] += dy[jj] * wr_[kx];
                dwa[ky * K + kx] += xa[t] * dy[jj];
              }
              asm volatile("" : "+v"(dwa[ky * K + kx]));
            }
#pragma unroll
            for (int t = 0; t < HW7; ++t) asm volatile("" : "+v"(dx[t]));
            __builtin_amdgcn_sched_barrier(0);
          }
          (void)jhi;
#pragma unroll
          for (int t = 0; t < HW7; ++t) xq[t] = xp[t];
#pragma unroll
          for (int t = 0; t < HW7; ++t) {
            const float x0 = X::lo(xq[t]), x1 = X::hi(xq[t]);
            float v0 = cw_act_bwd(dx[t][0], x0 * sc0 + sh0, in_relu, AM);
            float v1 = cw_act_bwd(dx[t][1], x1 * sc1 + sh1, in_relu, AM);
            v0 = ch0_ok ? v0 : 0.f;
            v1 = ch1_ok ? v1 : 0.f;
            const pair_t o = X::pack(v0, v1);
            v0 = X::lo(o); v1 = X::hi(o);
            s0a += v0; s0b += v1;
            s1a += v0 * x0; s1b += v1 * x1;
            xp[t] = o;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    pn0 = n0; phd0 = hd0; ppmax = cur_pmaxx;
    nb = nnb; ty = nty;
    if (g.ring) { base += g.THd; if (base >= g.LH) base -= g.LH; }
  }
  __syncthreads();
  if (pn0 >= 0) store_h(pn0, phd0, ppmax);

  {   // weight-gradient flush through this wave's own plane, see k_dwb_cw
    constexpr int G = 14, NV = 2 * KK;
    float* red = reinterpret_cast<float*>(s_dy + wv * g.plane);
    float* red2 = red + 64 * (G + 1);
    float* drow = dwp ? dwp + ((long)worker * g.C + ch) * KK : nullptr;
    const int nvalid = ch1_ok ? NV : (ch0_ok ? KK : 0);
    const int rq = lane / G, rv = lane - rq * G;
#pragma unroll
    for (int r0 = 0; r0 < NV; r0 += G) {
#pragma unroll
      for (int v = 0; v < G; ++v)
        if (r0 + v < NV) red[lane * (G + 1) + v] = (r0 + v < KK) ? dwa[(r0 + v) % KK][0] : dwa[(r0 + v) % KK][1];
      __builtin_amdgcn_wave_barrier();
      float part = 0.f;
      if (lane < 4 * G) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) part += red[(rq * 16 + i) * (G + 1) + rv];
        red2[lane] = part;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < G && drow && r0 + lane < nvalid) drow[r0 + lane] = ((red2[lane] + red2[G + lane]) + red2[2 * G + lane]) + red2[3 * G + lane];
      __builtin_amdgcn_wave_barrier();
    }
  }
  s0a = cw_wave_sum63(s0a); s0b = cw_wave_sum63(s0b); s1a = cw_wave_sum63(s1a); s1b = cw_wave_sum63(s1b);
  if (lane == 63 && stats) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = ch + e;
      if (c < g.C) {
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

// ---------------------------------------------------------------------------------------------------------------- forward
//   y = dwconv(act(x*in_scale+in_shift)),  stats: sum y, sum y^2 (of the stored values)
template <typename T, int K, int AM, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void k_dwf_cw(const T* __restrict__ x, long xss, const float* __restrict__ in_scale,
                                                     const float* __restrict__ in_shift, int in_relu, const float* __restrict__ w,
                                                     int ldw, T* __restrict__ y, long yss, float* __restrict__ stats, int stat_ld,
                                                     int stat_rows, CwGeom g) {
  typedef Cw<T> X;
  typedef typename X::pair_t pair_t;
  typedef typename X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, SW = 7, IWN = SW + K - 1, NT = NW * 64, CGS = NW / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* s_in = reinterpret_cast<f32x2*>(smem);                     // [NW pairs][plane]: activated input window, fp32
  pair_t* s_y = reinterpret_cast<pair_t*>(s_in + NW * g.plane);      // [NW pairs][TPIXp]: the tile's output
  float* s_cf = reinterpret_cast<float*>(s_y + NW * g.TPIXp);        // [2][16] scale / shift of the slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int slab, worker, half;
  if (!cw_block<NW>(g, slab, worker, half)) return;   // surplus block of the padded half-slab grid (whole workgroup, before any barrier)
  const int c_base = slab * 16;
  const int ch = c_base + 2 * (wv + 4 * half);
  const int cpad = (g.C + 7) & ~7;
  const int cgl = CGS == 2 ? (tid & 1) : 0;   // channel group inside the workgroup's planes
  const int cg = cgl + half;                   // channel group inside the slab
  const bool cg_ok = c_base + cg * 8 < cpad;

  for (int i = tid; i < NW * g.plane; i += NT) s_in[i] = f32x2{0.f, 0.f};
  if (tid < 32) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    s_cf[tid] = (in_scale && c < cpad) ? (v == 0 ? in_scale[c] : in_shift[c]) : (v == 0 ? 1.f : 0.f);
  }
  const bool ch0_ok = ch < g.C, ch1_ok = ch + 1 < g.C;
  const float* wp = w + ch;   // taps of the pair as wave-uniform scalars, see k_dwb_cw
  unsigned ld4 = (unsigned)ldw * 4u;

  CwSlots sl;
  cw_decode<P, NT, CGS>(sl, g, tid, cg);
  const int ipi = g.TH * g.ns;
  const int it_im = lane / ipi, it_rem = lane % ipi;
  const int it_r = it_rem / g.ns, it_j = it_rem % g.ns;
  const bool it_ok = lane < g.NI * ipi;
  const int pix0 = (it_im * g.TH + it_r) * g.W + SW * it_j;
  const unsigned in_addr0 = cw_lds_addr(s_in + wv * g.plane + it_im * g.RH * g.LWp + SW * it_j);

  float sa = 0.f, sb = 0.f, qa = 0.f, qb = 0.f;
  piece_t pfx[2];
  unsigned pxmask = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) X::zero(pfx[i]);

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_x = (long)slab * xss, slab_y = (long)slab * yss;

  auto issue = [&](int n0, int ho0) {   // branch-free, see k_dwb_cw
    const int hi_s = g.ring ? ho0 + P : 0;
    const long px = ((long)n0 * g.H + hi_s) * g.W * 16;
    pxmask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool okx = sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N && hi_s + sl.rr[i] < g.H;
      X::load(pfx[i], x + slab_x + (okx ? px + sl.goff[i] : 0));
      pxmask |= okx ? 1u << i : 0u;
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
        put_in(pfx[i], (pxmask >> i) & 1u, s_in + (cgl * 4) * g.plane + sl.dyo[i] + slot * g.LWp);
      }
    }
  };
  auto halo_sync = [&](int n0, int ho0) {
    const int npc = 2 * P * g.W * CGS;
    for (int p = tid; p < npc; p += NT) {
      const int col = (p / CGS) % g.W, wr = (p / CGS) / g.W;
      const int hi = ho0 - P + wr;
      piece_t a;
      X::zero(a);
      const bool ok = cg_ok && hi >= 0 && hi < g.H && n0 < g.N;
      if (ok) X::load(a, x + slab_x + (((long)n0 * g.H + hi) * g.W + col) * 16 + cg * 8);
      put_in(a, ok, s_in + (cgl * 4) * g.plane + wr * g.LWp + col + P);
    }
  };
  auto store_y = [&](int n0, int ho0) {
    const long py = ((long)n0 * g.H + ho0) * g.W * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N && ho0 + sl.rr[i] < g.H) {
        piece_t v;
        const pair_t* sy_ = s_y + (cgl * 4) * g.TPIXp + sl.pp[i];
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
    asm volatile("" : "+v"(sl.goff[0]), "+v"(sl.goff[1]), "+v"(sl.pp[0]), "+v"(sl.pp[1]), "+v"(sl.dyo[0]), "+v"(sl.dyo[1]));   // see k_dwb_cw
    __syncthreads();   // (A) previous tile consumed, its output complete in s_y
    X::touch(pfx[0]); X::touch(pfx[1]);   // see k_dwb_cw
    if (pn0 >= 0) store_y(pn0, pho0);
    commit(base);
    if (fresh) halo_sync(n0, ho0);
    __syncthreads();   // (B)
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.TH);

    asm volatile("" : "+s"(ld4));   // see k_dwb_cw
    if (it_ok && n0 + it_im < g.N && ho0 + it_r < g.H && ch < cpad) {
      f32x2 acc[SW];
#pragma unroll
      for (int t = 0; t < SW; ++t) acc[t] = f32x2{0.f, 0.f};
      f32x2 inb[2][IWN], wb[2][K];   // two operand buffers: row ky + 1 is in flight while row ky is multiplied
      auto row_addr = [&](int ky) {
        int slot = it_r + ky + base;
        if (slot >= g.LH) slot -= g.LH;
        return in_addr0 + (unsigned)(slot * g.LWp) * 8u;
      };
      cw_row_issue<K, IWN>(wb[0], inb[0], wp, 0u, ld4, row_addr(0));
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int cur = ky & 1;
        cw_row_wait();
        if (ky + 1 < K) {
          cw_row_issue<K, IWN>(wb[cur ^ 1], inb[cur ^ 1], wp, (unsigned)((ky + 1) * K) * ld4, ld4, row_addr(ky + 1));
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int t = 0; t < SW; ++t) acc[t] += inb[cur][t + kx] * wb[cur][kx];
        }
#pragma unroll
        for (int t = 0; t < SW; ++t) asm volatile("" : "+v"(acc[t]));   // this row's FMAs are done before the next row's operands
        __builtin_amdgcn_sched_barrier(0);
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
    sa = cw_wave_sum63(sa); sb = cw_wave_sum63(sb); qa = cw_wave_sum63(qa); qb = cw_wave_sum63(qb);
    if (lane == 63) {
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


template <typename T, int K>
static int cw_launch_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                         const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                         float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  if (!cw_geometry(g, N, H, W, C, K)) return -1;
  const int nw = cw_nw();
  const size_t lds = cw_lds<T>(g, nw);
  if (lds > max_lds_bytes()) return -1;
  // waves per SIMD the instances are compiled for: half-slab workgroups 3 (k = 3: 139 registers) or 2; whole-slab workgroups 2
  constexpr int WPS4 = K == 3 ? 3 : 2;
#define CW_BWD(KERN, AMV, NWV, WPSV)                                                                                        \
  {                                                                                                                         \
    auto kern = KERN<T, K, AMV, NWV, WPSV, (K == 7)>;                                                                                 \
    cw_workers(g, resident_per_cu(kern, NWV * 64, lds), (stats || dw) ? part_rows : 0, NWV);                                \
    hipLaunchKernelGGL(kern, dim3(cw_grid(g, NWV)), dim3(NWV * 64), lds, st, (const T*)gup, gss, (const T*)yraw, yrss, c1, c2, \
                       c3, (const T*)x, xss, sc, sh, relu, w, ldw, (T*)h, hss, dw ? dw_ws : nullptr, stats, stat_ld, part_rows, g); \
  }
  if (nw == 4) {
    if (relu == ACT_RELU6) CW_BWD(k_dwb_cw, ACT_RELU6, 4, WPS4) else if (relu == ACT_SWISH) CW_BWD(k_dwb_cw, ACT_SWISH, 4, WPS4) else CW_BWD(k_dwb_cw, 0, 4, WPS4)
  } else {
    if (relu == ACT_RELU6) CW_BWD(k_dwb_cw, ACT_RELU6, 8, 2) else if (relu == ACT_SWISH) CW_BWD(k_dwb_cw, ACT_SWISH, 8, 2) else CW_BWD(k_dwb_cw, 0, 8, 2)
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
  const int nw = cw_nw();
  const size_t lds = cw_lds<T>(g, nw);
  if (lds > max_lds_bytes()) return -1;
#define CW_FWD(AMV, NWV)                                                                                                    \
  {                                                                                                                         \
    auto kern = k_dwf_cw<T, K, AMV, NWV, 4>;                                                                                \
    cw_workers(g, resident_per_cu(kern, NWV * 64, lds), stats ? stat_rows : 0, NWV);                                        \
    hipLaunchKernelGGL(kern, dim3(cw_grid(g, NWV)), dim3(NWV * 64), lds, st, (const T*)x, xss, sc, sh, relu, w, ldw, (T*)y, yss, \
                       stats, stat_ld, stat_rows, g);                                                                       \
  }
  if (nw == 4) {
    if (relu == ACT_RELU6) CW_FWD(ACT_RELU6, 4) else if (relu == ACT_SWISH) CW_FWD(ACT_SWISH, 4) else CW_FWD(0, 4)
  } else {
    if (relu == ACT_RELU6) CW_FWD(ACT_RELU6, 8) else if (relu == ACT_SWISH) CW_FWD(ACT_SWISH, 8) else CW_FWD(0, 8)
  }
#undef CW_FWD
  return check_launch("dwconv_fwd(cw)");
}

template <typename T, int K>
static int cw2_launch_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                          const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                          float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  if (!cw2_geometry(g, N, H, W, C, K)) return -1;
  const size_t lds = cw_lds<T>(g, 4);
  if (lds > max_lds_bytes()) return -1;
#define CW2_BWD(AMV)                                                                                                        \
  {                                                                                                                         \
    auto kern = k_dwb_cw2<T, K, AMV, 4, 2>;                                                                                 \
    cw_workers(g, resident_per_cu(kern, 256, lds), (stats || dw) ? part_rows : 0, 4);                                       \
    hipLaunchKernelGGL(kern, dim3(cw_grid(g, 4)), dim3(256), lds, st, (const T*)gup, gss, (const T*)yraw, yrss, c1, c2,     \
                       c3, (const T*)x, xss, sc, sh, relu, w, ldw, (T*)h, hss, dw ? dw_ws : nullptr, stats, stat_ld, part_rows, g); \
  }
  if (relu == ACT_RELU6) CW2_BWD(ACT_RELU6) else if (relu == ACT_SWISH) CW2_BWD(ACT_SWISH) else CW2_BWD(0)
#undef CW2_BWD
  if (int rc = check_launch("dwconv_bwd(cw2)")) return rc;
  if (dw) return reduce_parts(dw_ws, (long)C * K * K, g.nworkers, (long)C * K * K, dw, C * K * K, 0, 1, st);
  return 0;
}


// -1: not one of this file's cases (the caller continues with the tile kernels of dwconv.hip); otherwise the launch status
int dwconv_cw_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                  const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                  float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k, int stride,
                  int dtype, hipStream_t st) {
  if (!(cw_mode() & 1) || gss == 0 || xss == 0 || hss == 0 || (yraw && yrss == 0) || ldw < ((C + 7) & ~7)) return -1;
  if (stride == 2) {
    if (!(cw_mode() & 4)) return -1;
#define CW_B2(TT, KV) return cw2_launch_bwd<TT, KV>(gup, gss, yraw, yrss, c1, c2, c3, x, xss, sc, sh, relu, w, ldw, h, hss, dw, stats, stat_ld, part_rows, dw_ws, N, H, W, C, st)
    if (dtype == DT_F32) {
      if (k == 3) CW_B2(float, 3); if (k == 5) CW_B2(float, 5); if (k == 7) CW_B2(float, 7);
    } else {
      if (k == 3) CW_B2(bf16_t, 3); if (k == 5) CW_B2(bf16_t, 5); if (k == 7) CW_B2(bf16_t, 7);
    }
#undef CW_B2
    return -1;
  }
  if (stride != 1) return -1;
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
  if (!(cw_mode() & 2) || xss == 0 || yss == 0 || ldw < ((C + 7) & ~7)) return -1;
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

#if CW_TIMING
extern "C" int atomnas_debug_cw_timing(unsigned long long* out8, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(atomnas::g_cw_timing), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(atomnas::g_cw_timing), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif

// 1 when atomnas_dwconv_fwd (dir = 0) / atomnas_dwconv_bwd (dir = 1) take the channel-pair-per-wave kernels of this file for the
// shape (slab-major tensors, stride 1), 0 when they take the tile kernels of dwconv.hip.  Tests and launch-geometry tools only.
extern "C" int atomnas_dwconv_cw_supported(int N, int H, int W, int C, int k, int stride, int dtype, int dir) {
  using namespace atomnas;
  if (!(k == 3 || k == 5 || k == 7)) return 0;
  CwGeom g;
  if (stride == 2) {
    if (!(cw_mode() & (dir ? 4 : 8)) || dir == 0) return 0;   // stride 2: backward only so far
    if (!cw2_geometry(g, N, H, W, C, k)) return 0;
    const size_t lds = dtype == DT_F32 ? cw_lds<float>(g, 4) : cw_lds<bf16_t>(g, 4);
    return lds <= max_lds_bytes() ? 1 : 0;
  }
  if (stride != 1 || !(cw_mode() & (dir ? 1 : 2))) return 0;
  if (!cw_geometry(g, N, H, W, C, k)) return 0;
  const size_t lds = dtype == DT_F32 ? cw_lds<float>(g, cw_nw()) : cw_lds<bf16_t>(g, cw_nw());
  return lds <= max_lds_bytes() ? 1 : 0;
}

// Round-4 experiment, NOT part of the product library (built by tools/build_xdw_experiment.sh only): the channel-pair depthwise
// backward with its input operand E = x We^T recomputed on chip ("E-elimination"; measured slower than
// reading E).  Moved out of csrc/dwconv_cw.hip in round 5; entry points declared in xdw_experimental.h / used by xdw_fused.hip.
#include "../dwconv_cw.h"
#include "xdw_internal.h"

#undef CW_TIMING
#define CW_TIMING 0
#undef CWMARK
#define CWMARK(i)

namespace atomnas {

// ------------------------------------------------------------------------------------------- backward, expand recomputed
// k_dwb_cw with its input operand recomputed on chip ("E-elimination", see xdw.hip): the raw expand output E = xin W1^T of the
// tile's pixels comes out of an MFMA stage (weights of the slab as the A operand, 16-pixel groups of the narrow block input xin
// as the B operand straight from global memory, prefetched behind the tap loop like the other streams) instead of a fourth HBM
// stream, in fp32 (it used to be read back rounded to bf16).  Half-slab workgroups of 4 waves: the MFMA tile covers the 16
// channels of the slab, the workgroup keeps its 8.
//   dYraw = c1*g + c2*yraw + c3;  e = xin W1^T;  h = dwconv^T(dYraw) * act'(e*in_scale+in_shift),
//   dW += corr(act(e*in_scale+in_shift), dYraw),  stats: sum h, sum h*e
template <int K, int AM, int KC, int WPS, bool PF>
__global__ __launch_bounds__(256, WPS) void k_xdwb(const bf16_t* __restrict__ gup, long gss, const bf16_t* __restrict__ yraw, long yrss,
                                                   const float* __restrict__ c1, const float* __restrict__ c2p,
                                                   const float* __restrict__ c3, const bf16_t* __restrict__ xin, int ldx, int inp,
                                                   const bf16_t* __restrict__ wexp, int ldwe,
                                                   const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                   int in_relu, const float* __restrict__ w, int ldw, bf16_t* __restrict__ h, long hss,
                                                   float* __restrict__ dwp, float* __restrict__ stats, int stat_ld, int stat_rows,
                                                   CwGeom g) {
  typedef bf16_t T;
  constexpr int NW = 4, GPW = 7;
  constexpr bool LATEB = (K == 7);
  typedef Cw<T> X;
  typedef typename X::pair_t pair_t;
  typedef typename X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, KK = K * K, SW = 7, DWN = SW + K - 1, NT = NW * 64, CGS = NW / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* s_dy = reinterpret_cast<f32x2*>(smem);                    // [NW pairs][plane]: dYraw window, fp32
  f32x2* s_xe = s_dy + NW * g.plane;                                 // [NW pairs][TPIXp]: raw expand output of the tile, fp32
  pair_t* s_x = reinterpret_cast<pair_t*>(s_xe + NW * g.TPIXp);       // [NW pairs][TPIXp]: h of the tile (bf16 pairs)
  float* s_cf = reinterpret_cast<float*>(s_x + NW * g.TPIXp);        // [3][16] BN-backward coefficients of the slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);           // channel pair of this wave (wave-uniform)
  const int mq_ = lane >> 4, mj_ = lane & 15;                         // MFMA lane coordinates
  int slab, worker, half;
  {
    // blocks b and b + 8 (the same XCD under round-robin placement) are the two halves of one (slab, worker) unit.  Every XCD owns a
    // CONTIGUOUS range of the worker-major unit list, i.e. all slabs of a few workers -- the same pixel tiles of xin -- so its L2
    // fetches those tiles once (a worker whose slabs straddle two XCDs is fetched twice).
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    half = li & 1;
    const int units = g.nworkers * g.nslabs, upx = (units + 7) >> 3;
    const int pidx = li >> 1;
    const int unit = xcd * upx + pidx;
    if (pidx >= upx || unit >= units) return;   // surplus block (whole workgroup, before any barrier)
    worker = unit / g.nslabs;
    slab = unit - worker * g.nslabs;
  }
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

  piece_t pfg[2], pfy[2];
  unsigned pfmask = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) { X::zero(pfg[i]); X::zero(pfy[i]); }
  // expand weights of the slab as MFMA A fragments (row = channel c_base + mj), fetched with the pixels of every tile (L1 hits)
  bf16x8 afr[KC];
  bf16x8 bfr[GPW][KC];   // pixels of xin: group gi of this wave = tile pixels 16 (4 gi + wv) .. + 15

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_g = (long)slab * gss, slab_y = (long)slab * yrss, slab_h = (long)slab * hss;

  // issue the HBM loads of a tile (n0 = first image, hi0 = first row): dY rows [ho_s, ho_s + TH), input rows [hi0, hi0 + TH)
  // Branch-free: a piece that does not exist (image / row beyond the tensor, channel group beyond C) reads the first piece of the
  // slab instead and is masked at the commit.  (With the loads under per-lane branches hipcc cannot prove at the loop back-edge
  // that they were waited for, and puts an s_waitcnt vmcnt(0) in front of the next tile's loads: that wait also covers the h
  // stores issued just before -- 20 to 30 % of the kernel in the first measurements.)
  auto issue = [&](int n0, int hi0) {
    const int ho_s = g.ring ? hi0 + P : 0;
    const long pg = ((long)n0 * g.H + ho_s) * g.W * 16;
    pfmask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool in_n = sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N;
      const bool okg = in_n && ho_s + sl.rr[i] < g.H;
      const long og = okg ? pg + sl.goff[i] : 0;
      X::load(pfg[i], gup + slab_g + og);
      if (yraw) X::load(pfy[i], yraw + slab_y + og);
      pfmask |= okg ? 1u << i : 0u;
    }
  };
  // the tile's pixels of xin (contiguous: whole rows / whole images); beyond the tensor: the first bytes of xin, zeroed.
  // k = 7 (LATEB): issued behind the tap loop instead of in front of it -- its operand buffers and the weight-gradient
  // accumulators leave no registers for the fragments; the activation epilogue, barrier (A), the h stores and the dY commit
  // cover the (L2) latency.
  auto issue_b = [&](int n0, int hi0) {
    // opaque per tile: the fragment addresses are formed here (hoisted out of the tile loop they stay live across the tap rows)
    int mj = mj_, mq = mq_;
    asm volatile("" : "+v"(mj), "+v"(mq));
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) afr[kc] = *reinterpret_cast<const bf16x8*>(wexp + (long)(c_base + mj) * ldwe + kc * 32 + 8 * mq);
    const long base_px = ((long)n0 * g.H + hi0) * g.W;
    const long lim_l = g.NI == 1 ? (long)(g.H - hi0) * g.W : (long)(g.N - n0) * g.H * g.W;
    const int lim = (int)(lim_l < g.TPIX ? lim_l : g.TPIX);
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int pp = (gi * 4 + wv) * 16 + mj;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int kk = kc * 32 + 8 * mq;
        // beyond the tensor / beyond inp: the first bytes of xin -- finite values that meet zero weight columns or are never read
        const bool ok = pp < lim && kk < inp;
        bfr[gi][kc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xin + (ok ? (base_px + pp) * ldx + kk : 0)));
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
        put_dy(pfg[i], pfy[i], (pfmask >> i) & 1u, s_dy + (cgl * 4) * g.plane + sl.dyo[i] + slot * g.LWp);
      }
    }
    // MFMA stage: raw expand output of the tile's pixels for the 16 channels of the slab; the lanes that hold this workgroup's
    // half (channels 4 mq .. 4 mq + 3 with mq >> 1 == half) write their two channel pairs
    int mj = mj_, mq = mq_;
    asm volatile("" : "+v"(mj), "+v"(mq));
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if ((gi * 4 + wv) * 16 < g.TPIX) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], bfr[gi][kc], acc, 0, 0, 0);
        const int pp = (gi * 4 + wv) * 16 + mj;
        if ((mq >> 1) == half && pp < g.TPIX) {
          f32x2* d = s_xe + (2 * (mq & 1)) * g.TPIXp + pp;
          d[0] = f32x2{acc[0], acc[1]};
          d[g.TPIXp] = f32x2{acc[2], acc[3]};
        }
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
  if (tile < t_end) { issue(nb * g.NI, ty * g.TH); issue_b(nb * g.NI, ty * g.TH); }
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
    for (int i = 0; i < 2; ++i) { X::touch(pfg[i]); X::touch(pfy[i]); }
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) asm volatile("" ::"v"(bfr[gi][kc]));
    if (pn0 >= 0) store_h(pn0, phi0);
    commit(base);
    if (fresh) halo_sync(n0, hi0);
    CWMARK(1)
    __syncthreads();   // (B) window and pixel planes complete
    CWMARK(2)
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) {
      issue(nnb * g.NI, nty * g.TH);
      if (!LATEB) issue_b(nnb * g.NI, nty * g.TH);
    }
    CWMARK(3)

    // opaque per tile: the k^2 tap offsets are formed in their tap rows instead of being hoisted out of the tile loop (they spill)
    asm volatile("" : "+s"(ld4));
    if (it_ok && n0 + it_im < g.N && hi0 + it_r < g.H && ch < cpad) {
      pair_t* xp = s_x + wv * g.TPIXp + pix0;
      const f32x2* ep = s_xe + wv * g.TPIXp + pix0;
      f32x2 xq[SW];
      f32x2 xa[SW], dx[SW];
#pragma unroll
      for (int t = 0; t < SW; ++t) xq[t] = ep[t];
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        xa[t] = f32x2{cw_act(xq[t][0] * sc0 + sh0, in_relu, AM), cw_act(xq[t][1] * sc1 + sh1, in_relu, AM)};
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
      // epilogue: activation backward of the producer, rounding, statistics; h goes to its own plane
#pragma unroll
      for (int t = 0; t < SW; ++t) xq[t] = ep[t];   // read again: the raw values are not kept in registers across the tap rows
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        const float x0 = xq[t][0], x1 = xq[t][1];
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
    // unconditional (the last tile fetches its own pixels again): under a branch the old fragments would stay live across the tap rows
    if (LATEB) { const bool more = tile + 1 < t_end; issue_b(more ? nnb * g.NI : n0, more ? nty * g.TH : hi0); }
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


template <int K, int KC>
static int xdw_launch_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                          const void* xin, int ldx, int inp, const void* wexp, int ldwe, const float* sc, const float* sh, int relu,
                          const float* w, int ldw, void* h, long hss, float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws,
                          int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  if (!cw_geometry(g, N, H, W, C, K)) return -1;
  const size_t lds = (size_t)4 * g.plane * sizeof(f32x2) + (size_t)4 * g.TPIXp * (sizeof(f32x2) + sizeof(unsigned)) + 48 * sizeof(float);
  if (lds > max_lds_bytes()) return -1;
  constexpr int WPSV = (K == 3 && KC == 1) ? 3 : 2;
#define XDW_BWD(AMV)                                                                                                          \
  {                                                                                                                           \
    auto kern = k_xdwb<K, AMV, KC, WPSV, (K == 7)>;                                                                           \
    cw_workers(g, resident_per_cu(kern, 256, lds), (stats || dw) ? part_rows : 0, 4);                                         \
    hipLaunchKernelGGL(kern, dim3((unsigned)(((g.nworkers * g.nslabs + 7) / 8) * 16)), dim3(256), lds, st, (const bf16_t*)gup, gss, (const bf16_t*)yraw, yrss, c1, c2, c3, \
                       (const bf16_t*)xin, ldx, inp, (const bf16_t*)wexp, ldwe, sc, sh, relu, w, ldw, (bf16_t*)h, hss,       \
                       dw ? dw_ws : nullptr, stats, stat_ld, part_rows, g);                                                   \
  }
  if (relu == ACT_RELU6) XDW_BWD(ACT_RELU6) else if (relu == ACT_SWISH) XDW_BWD(ACT_SWISH) else XDW_BWD(0)
#undef XDW_BWD
  if (int rc = check_launch("xdw_bwd")) return rc;
  if (dw) return reduce_parts(dw_ws, (long)C * K * K, g.nworkers, (long)C * K * K, dw, C * K * K, 0, 1, st);
  return 0;
}

// the fused expand + depthwise backward (xdw.hip's entry point atomnas_xdw_bwd): -1 when the shape has no instance
int xdw_cw_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3, const void* xin,
               int ldx, int inp, const void* wexp, int ldwe, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h,
               long hss, float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k,
               hipStream_t st) {
  if (gss == 0 || hss == 0 || (yraw && yrss == 0) || ldw < ((C + 7) & ~7) || C % 16 || inp > 64) return -1;
  const int kc = (inp + 31) / 32;
#define XDW_B(KV, KCV) \
  if (k == KV && kc == KCV) return xdw_launch_bwd<KV, KCV>(gup, gss, yraw, yrss, c1, c2, c3, xin, ldx, inp, wexp, ldwe, sc, sh, relu, w, ldw, h, hss, dw, stats, stat_ld, part_rows, dw_ws, N, H, W, C, st);
  XDW_B(3, 1) XDW_B(5, 1) XDW_B(7, 1) XDW_B(3, 2) XDW_B(5, 2) XDW_B(7, 2)
#undef XDW_B
  return -1;
}
// 1 when xdw_cw_bwd has an instance for the shape
int xdw_cw_bwd_supported(int N, int H, int W, int C, int k) {
  CwGeom g;
  if (!(k == 3 || k == 5 || k == 7) || C % 16 || !cw_geometry(g, N, H, W, C, k)) return 0;
  const size_t lds = (size_t)4 * g.plane * sizeof(f32x2) + (size_t)4 * g.TPIXp * (sizeof(f32x2) + sizeof(unsigned)) + 48 * sizeof(float);
  return lds <= max_lds_bytes() ? 1 : 0;
}


}  // namespace atomnas

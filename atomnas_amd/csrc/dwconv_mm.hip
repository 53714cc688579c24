// Depthwise k x k convolution with the tap arithmetic on the matrix cores (gfx950), bf16 storage, stride 1, slab-major tensors.
//
// Same operation and the same C-ABI entry points as dwconv.hip / dwconv_cw.hip (models/mobilenet_base.py:330-336: the depthwise
// ConvBNReLU of the atomic block).  Round-3/4 measurements: the channel-pair kernels of
// dwconv_cw.hip are bound by VALU issue for k = 5 and k = 7 (343 packed FMAs per 448 pixel pairs at k = 7 against ~390 other vector
// instructions; 2.3 TB/s where the k = 3 instance streams 3.3).  A depthwise convolution has no contraction over channels, but per
// CHANNEL it is a small dense contraction over the k x k window, and with a constant operand that is a banded (Toeplitz) matrix of
// the taps it maps onto v_mfma_f32_16x16x32_f16:
//
//     one MFMA "tile" = 2 output rows x 8 output columns of one channel (M = 16 outputs);  its input patch is (2 + k - 1) rows x 16
//     columns of the activated input window; the patch is cut into k-blocks of 8 consecutive columns of one row (16 bytes of fp16 in
//     LDS, 16-byte aligned because tiles start at multiples of 8 columns), four k-blocks = two patch rows per MFMA (K = 32):
//         D[m = (orow, ocol)][n = tile] += sum_k A[m][k = (prow, pcol)] * B[k][n],    A[m][k] = w[prow - orow][pcol - ocol] or 0
//     A (the Toeplitz operand, per channel and pair of patch rows) is built once per wave from the taps and stays in registers: (k + 1) / 2
//     fragments of 4 registers per channel; B is the data: lane (n, q) reads 16 bytes of tile n's patch.  N = 16 tiles of the pixel
//     tile are processed per MFMA.  k = 7: 4 MFMAs per 256 outputs of a channel (about 64 matrix-pipe cycles) where the packed-FMA
//     rows need 392 VALU cycles; 4 LDS reads of 1 KB where they need 26 of 512 B.
//
// Everything around the tap arithmetic is dwconv_cw.hip's forward kernel: half-slab workgroups of 4 waves, a wave owns one channel
// pair, full-width row-ring tiles (no halo re-reads), branch-free prefetch of the next tile, output staged through LDS as packed channel
// pairs so that HBM sees 16-byte accesses only, statistics as per-lane partials added in a fixed order.
//
// Numerics: the operands of the MFMA are fp16 (the activated input act(x * scale + shift), computed in fp32 from the bf16 tensor, and
// the taps; both rounded to nearest even, the input clamped to the fp16 range), products are exact in fp32 and accumulated in fp32:
// 2^-12 relative per operand, an eighth of the rounding of the bf16 OUTPUT (2^-9), which stays the dominant term.  bf16 operands
// would have the range but 2^-9 per operand; the inputs here are BatchNorm outputs behind an activation (O(1), not driven to zero by
// the L1 penalty, which acts on the BatchNorm AFTER this convolution).  The fp32 parity mode never takes these kernels.
// oracle/atomnas_oracle.py restates the two roundings (Bf16Storage(dw_fp16=True)).  Bit-reproducible; no atomics.
#include "dwconv_cw.h"

namespace atomnas {

typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int MM_MAXG = 4;   // MFMA tile groups (16 tiles each) of a pixel tile: the per-lane decode of a group lives in registers

// extra geometry of the matrix-core kernels on top of CwGeom (tiles, workers and the staging slots are dwconv_cw.hip's)
struct MmGeom {
  int nrp, ncb, ntl, ngroups;   // row pairs / 8-column blocks per image of the pixel tile, MFMA tiles per pixel tile, groups of 16
  int nkb, xwp, xplane, steps;  // backward: 8-column blocks of a window row, row pitch / plane elements of the activated-input copies,
                                // groups of four rows of the pixel tile (weight gradient: nkb MFMAs per group)
};

__device__ __forceinline__ float mm_clamp16(float a) { return __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); }

// ---------------------------------------------------------------------------------------------------------------- forward
//   y = dwconv(act(x*in_scale+in_shift)),  stats: sum y, sum y^2 (of the stored values)
template <int K, int AM, int WPS>
__global__ __launch_bounds__(256, WPS) void k_dwf_mm(const bf16_t* __restrict__ x, long xss, const float* __restrict__ in_scale,
                                                  const float* __restrict__ in_shift, int in_relu, const float* __restrict__ w, int ldw,
                                                  bf16_t* __restrict__ y, long yss, float* __restrict__ stats, int stat_ld, int stat_rows,
                                                  CwGeom g, MmGeom mg) {
  typedef Cw<bf16_t> X;
  typedef X::pair_t pair_t;
  typedef X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, KK = K * K, NW = 4, NT = 256, CGS = 1, NJ = (K + 1) / 2;   // NJ MFMAs (2 patch rows each) per tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f16_t* s_in = reinterpret_cast<f16_t*>(smem);                          // [8 channels][plane]: activated input window, fp16
  pair_t* s_y = reinterpret_cast<pair_t*>(s_in + 8 * g.plane);            // [4 pairs][TPIXp]: the tile's output
  float* s_cf = reinterpret_cast<float*>(s_y + NW * g.TPIXp);             // [2][16] scale / shift of the slab
  float* s_w = s_cf + 32;                                                 // [8][KK] taps of the workgroup's channels

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int slab, worker, half;
  if (!cw_block<NW>(g, slab, worker, half)) return;   // surplus block of the padded half-slab grid (whole workgroup, before any barrier)
  const int c_base = slab * 16;
  const int ch = c_base + 2 * (wv + 4 * half);
  const int cpad = (g.C + 7) & ~7;
  const int cg = half;
  const bool cg_ok = c_base + cg * 8 < cpad;

  {   // halo columns / rows outside the image stay zero
    u32x4* z = reinterpret_cast<u32x4*>(s_in);
    for (int i = tid; i < g.plane; i += NT) z[i] = u32x4{0u, 0u, 0u, 0u};   // 8 planes x plane halves = plane 16-byte pieces
  }
  if (tid < 32) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    s_cf[tid] = (in_scale && c < cpad) ? (v == 0 ? in_scale[c] : in_shift[c]) : (v == 0 ? 1.f : 0.f);
  }
  for (int i = tid; i < 8 * KK; i += NT) {
    const int c = i / KK, t = i - c * KK, chn = c_base + 8 * half + c;
    s_w[i] = chn < cpad ? w[(long)t * ldw + chn] : 0.f;   // (taps of channels in [C, cpad) are whatever the table holds: their outputs are forced to zero)
  }
  const bool ch0_ok = ch < g.C, ch1_ok = ch + 1 < g.C;

  // Tile-independent decode of the two staging slots.  A tile's pieces (8 channels of a pixel) run over (image, row, column) and are
  // contiguous in the slab: piece tid + 256 i is 16 (tid + 256 i) elements behind the tile's first one, and the VALID pieces (rows inside
  // the image, images inside the batch) are a prefix -- one comparison per slot; only the window offset needs the decode.
  int x_wo[2];   // (im * RH + row) * LWp + col + P
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = tid + i * NT;
    const int col = pp % g.W, t2 = pp / g.W;
    const int rr = t2 % g.TH, im = t2 / g.TH;
    x_wo[i] = (im * g.RH + rr) * g.LWp + col + P;
  }

  // per-lane decode of the MFMA tile groups: lane (n = lane & 15, q = lane >> 4) of group G works on tile 16 G + n; as the B operand it
  // supplies patch row 2 j + (q >> 1), columns 8 (q & 1) .. + 7 of MFMA j, as the D operand it receives output row q >> 1, columns
  // 4 (q & 1) .. + 3 of the tile
  const int nl = lane & 15, q = lane >> 4;
  int t_row[MM_MAXG], t_im[MM_MAXG], t_pp[MM_MAXG], t_ao[MM_MAXG];
  unsigned t_cm[MM_MAXG];
  {
    const int per_im = mg.nrp * mg.ncb;
#pragma unroll
    for (int G = 0; G < MM_MAXG; ++G) {
      const int t = 16 * G + nl;
      const bool tv = t < mg.ntl;
      const int tc = tv ? t : 0;
      const int im = tc / per_im, rem = tc - im * per_im;
      const int rp = rem / mg.ncb, cb = rem - rp * mg.ncb;
      const int row = 2 * rp + (q >> 1), col = 8 * cb + 4 * (q & 1);
      t_row[G] = row;
      t_im[G] = im;
      t_pp[G] = (im * g.TH + row) * g.W + col;
      t_ao[G] = im * g.RH * g.LWp + 8 * cb + 8 * (q & 1);
      unsigned cm = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cm |= (tv && row < g.TH && col + i < g.W) ? 1u << i : 0u;
      t_cm[G] = cm;
    }
  }

  float sa = 0.f, sb = 0.f, qa = 0.f, qb = 0.f;
  piece_t pfx[2];
  int pxn = 0;   // valid pieces of the prefetched tile
#pragma unroll
  for (int i = 0; i < 2; ++i) X::zero(pfx[i]);

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_x = (long)slab * xss, slab_y = (long)slab * yss;

  auto issue = [&](int n0, int ho0) {   // branch-free: invalid pieces load the tile's (or, with no valid piece at all, the slab's) first one
    const int hi_s = g.ring ? ho0 + P : 0;
    const int rows_ok = g.H - hi_s < g.TH ? g.H - hi_s : g.TH, ims_ok = g.N - n0 < g.NI ? g.N - n0 : g.NI;
    pxn = cg_ok ? (g.ring ? rows_ok * g.W : ims_ok * g.TH * g.W) : 0;
    const bf16_t* src = x + slab_x + (pxn > 0 ? ((long)n0 * g.H + hi_s) * g.W * 16 : 0) + cg * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = tid + i * NT;
      X::load(pfx[i], src + (pp < pxn ? (unsigned)pp * 16u : 0u));
    }
  };
  // one piece (8 channels of a pixel) -> the 8 channel planes
  auto put_in = [&](const piece_t& p, bool ok, f16_t* d) {
    float q1[8], q2[8];
    VecIO<float, 8>::load(s_cf + cg * 8, q1);
    VecIO<float, 8>::load(s_cf + 16 + cg * 8, q2);
    const CwClamp cb = cw_clamp16_bounds(ok, in_relu, AM);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const pair_t xq = X::pair(p, qq);
      const f32x2 bn = __builtin_elementwise_fma(f32x2{X::lo(xq), X::hi(xq)}, f32x2{q1[2 * qq], q1[2 * qq + 1]}, f32x2{q2[2 * qq], q2[2 * qq + 1]});   // v_pk_fma_f32
      float a0 = bn[0], a1 = bn[1];
      a0 = cw_act_clamp16(a0, cb, AM); a1 = cw_act_clamp16(a1, cb, AM);   // activation, then into the fp16 range (0 for an invalid piece)
      // one packed conversion for the pair (v_cvt_pk_f16_f32), the halves stored by ds_write_b16 / ds_write_b16_d16_hi
      typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
      const f16x2_t pk = __builtin_convertvector(f32x2{a0, a1}, f16x2_t);
      d[(2 * qq) * g.plane] = (f16_t)pk[0];
      d[(2 * qq + 1) * g.plane] = (f16_t)pk[1];
    }
  };
  auto commit = [&](int base) {
    // window row of tile row rr: (ring ? 2 P : P) + rr + base, modulo LH (ring: rr = pp / W, the rows that wrap are a suffix of the pieces)
    const int rowbase = (g.ring ? 2 * P : P) + base;
    const int wrap_pp = g.ring ? (g.LH - rowbase) * g.W : 0x7fffffff;
    const int o0 = rowbase * g.LWp, o1 = o0 - g.LH * g.LWp;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = tid + i * NT;
      if (pp < g.TPIX) put_in(pfx[i], pp < pxn, s_in + x_wo[i] + (pp >= wrap_pp ? o1 : o0));
    }
  };
  auto halo_sync = [&](int n0, int ho0) {
    const int npc = 2 * P * g.W;
    for (int p = tid; p < npc; p += NT) {
      const int col = p % g.W, wr = p / g.W;
      const int hi = ho0 - P + wr;
      piece_t a;
      X::zero(a);
      const bool ok = cg_ok && hi >= 0 && hi < g.H && n0 < g.N;
      if (ok) X::load(a, x + slab_x + (((long)n0 * g.H + hi) * g.W + col) * 16 + cg * 8);
      put_in(a, ok, s_in + wr * g.LWp + col + P);
    }
  };
  auto store_y = [&](int n0, int ho0) {   // the tile's output pieces are contiguous in the slab as well
    const int rows_ok = g.H - ho0 < g.TH ? g.H - ho0 : g.TH, ims_ok = g.N - n0 < g.NI ? g.N - n0 : g.NI;
    const int nv = cg_ok ? (g.ring ? rows_ok * g.W : ims_ok * g.TH * g.W) : 0;
    bf16_t* dst = y + slab_y + ((long)n0 * g.H + ho0) * g.W * 16 + cg * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = tid + i * NT;
      if (pp < nv) {
        piece_t v;
        const pair_t* sy_ = s_y + pp;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) X::set_pair(v, qq, sy_[qq * g.TPIXp]);
        X::store(v, dst + (unsigned)pp * 16u);
      }
    }
  };

  int tile = t_beg;
  int nb = tile / g.tiles_y, ty = tile % g.tiles_y;
  if (tile < t_end) issue(nb * g.NI, ty * g.TH);

  // Toeplitz operands of the wave's two channels (after the first barrier below orders s_w): A[m = (orow, ocol)][k = (prow, pcol)];
  // lane (m = lane & 15, q) holds k-block 4 j + q of MFMA j: patch row 2 j + (q >> 1), patch columns 8 (q & 1) + e
  f16x8 ta0[NJ], ta1[NJ];
  __syncthreads();
  {
    const int orow = nl >> 3, ocol = nl & 7;
    const float* w0 = s_w + (2 * wv) * KK;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ky = 2 * j + (q >> 1) - orow;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kx = 8 * (q & 1) + e - ocol;
        const bool ok = ky >= 0 && ky < K && kx >= 0 && kx < K;
        const int idx = ok ? ky * K + kx : 0;
        const float v0 = w0[idx], v1 = w0[KK + idx];
        ta0[j][e] = (ok && ch0_ok) ? (f16_t)v0 : (f16_t)0.f;   // channels beyond C: zero operand, zero output
        ta1[j][e] = (ok && ch1_ok) ? (f16_t)v1 : (f16_t)0.f;
      }
    }
  }

  int base = 0;
  int pn0 = -1, pho0 = 0;
  const f16_t* in0 = s_in + (2 * wv) * g.plane;
  for (; tile < t_end; ++tile) {
    const int n0 = nb * g.NI, ho0 = ty * g.TH;
    const bool fresh = g.ring && (tile == t_beg || ty == 0);
    if (fresh) base = 0;
    __syncthreads();   // (A) previous tile consumed, its output complete in s_y
    X::touch(pfx[0]); X::touch(pfx[1]);   // see k_dwb_cw
    if (pn0 >= 0) store_y(pn0, pho0);
    commit(base);
    if (fresh) halo_sync(n0, ho0);
    __syncthreads();   // (B)
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.TH);

    if (ch < cpad) {
#pragma unroll
      for (int G = 0; G < MM_MAXG; ++G) {
        if (G < mg.ngroups) {
          f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
          f16x8 b0[NJ], b1[NJ];   // all B fragments of the group first: the reads overlap instead of a read -> MFMA chain per j
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            int r = t_row[G] + 2 * j + base;
            if (r >= g.LH) r -= g.LH;
            const f16_t* bp = in0 + t_ao[G] + r * g.LWp;
            b0[j] = *reinterpret_cast<const f16x8*>(bp);
            b1[j] = *reinterpret_cast<const f16x8*>(bp + g.plane);
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta0[j], b0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta1[j], b1[j], acc1, 0, 0, 0);
          }
          const bool ok = n0 + t_im[G] < g.N && ho0 + t_row[G] < g.H;
          const unsigned cm = ok ? t_cm[G] : 0u;
          pair_t* yp = s_y + wv * g.TPIXp + t_pp[G];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if ((cm >> i) & 1u) {
              const pair_t o = X::pack(acc0[i], acc1[i]);
              const float v0 = X::lo(o), v1 = X::hi(o);
              sa += v0; sb += v1; qa += v0 * v0; qb += v1 * v1;
              yp[i] = o;
            }
          }
        }
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

// ---------------------------------------------------------------------------------------------------------------- backward
//   dYraw = c1*g + c2*yraw + c3 (BN-backward of the BN behind the conv, on load; yraw == NULL: dYraw = g)
//   h = dwconv^T(dYraw) * act'(x*in_scale+in_shift),  dW += corr(act(x*in_scale+in_shift), dYraw),  stats: sum h, sum h*x
// Both halves of the tap arithmetic on the matrix cores, bf16 operands (dYraw has the range of a gradient: fp16 would flush it):
//   * input gradient: the forward's Toeplitz form with the taps flipped, data operand = the dYraw window rounded to bf16.
//   * weight gradient: dW[K-1-a][K-1-b] = sum_{r, c} xa[r][c - b] * dYw[r + a][c] over the window columns c of the tile's rows r
//     (dYw = the zero-padded dYraw window) is ONE product per 32 window elements for all k x k taps of BOTH channels of the pair:
//         D[m = (e, b)][n = (e', a)] += sum_k A[m][k = (r, c)] * B[k][n],     A = xa of channel e shifted by b columns,
//                                                                           B = window row r + a of channel e'
//     (the e == e' blocks are the gradients; 2 k <= 16 rows / columns).  B fragments are 16-byte aligned reads of the window; A
//     fragments start at column c - b, i.e. at any 2-byte offset: misaligned LDS reads work on gfx950 but take 64 LDS cycles
//     (tools/probe/ldsmis.hip), so the activated input is kept twice, the second copy shifted by one element, and a fragment is
//     four dword reads from the copy in which its start is dword aligned.  The accumulator lives across all tiles of the worker and
//     is the worker's partial row: no cross-lane reduction at the end (the MFMA summed over the pixels).
template <int K, int AM, int WPS, int MAXG>
__global__ __launch_bounds__(256, WPS) void k_dwb_mm(const bf16_t* __restrict__ gup, long gss, const bf16_t* __restrict__ yraw, long yrss,
                                                  const float* __restrict__ c1, const float* __restrict__ c2p, const float* __restrict__ c3,
                                                  const bf16_t* __restrict__ x, long xss, const float* __restrict__ in_scale,
                                                  const float* __restrict__ in_shift, int in_relu, const float* __restrict__ w, int ldw,
                                                  bf16_t* __restrict__ h, long hss, float* __restrict__ dwp, float* __restrict__ stats,
                                                  int stat_ld, int stat_rows, CwGeom g, MmGeom mg) {
  typedef Cw<bf16_t> X;
  typedef X::pair_t pair_t;
  typedef X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, KK = K * K, NW = 4, NT = 256, CGS = 1, NJ = (K + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bf16_t* s_dy = reinterpret_cast<bf16_t*>(smem);                         // [8 channels][plane]: dYraw window, bf16
  bf16_t* s_xa = s_dy + 8 * g.plane;                                      // [8 channels][2 copies][xplane]: activated input, left-padded by 8
  pair_t* s_x = reinterpret_cast<pair_t*>(s_xa + 16 * mg.xplane);         // [4 pairs][TPIXp]: raw input pixels, replaced by h in place
  float* s_cf = reinterpret_cast<float*>(s_x + NW * g.TPIXp);             // [3][16] BN-backward coefficients, [2][16] scale / shift
  float* s_w = s_cf + 80;                                                 // [8][KK] taps of the workgroup's channels

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int slab, worker, half;
  if (!cw_block<NW>(g, slab, worker, half)) return;
  const int c_base = slab * 16;
  const int ch = c_base + 2 * (wv + 4 * half);
  const int cpad = (g.C + 7) & ~7;
  const int cg = half;
  const bool cg_ok = c_base + cg * 8 < cpad;

  {   // halo columns / rows outside the image and the left padding of the activated input stay zero
    u32x4* z = reinterpret_cast<u32x4*>(s_dy);
    const int n16 = g.plane + 2 * mg.xplane;   // (8 plane + 16 xplane) elements of 2 bytes = that many 16-byte pieces
    for (int i = tid; i < n16; i += NT) z[i] = u32x4{0u, 0u, 0u, 0u};
  }
  if (tid < 48) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    const float* src = (v == 0) ? c1 : (v == 1 ? c2p : c3);
    s_cf[tid] = (c1 && src && (v == 0 || yraw) && c < cpad) ? src[c] : (v == 0 ? 1.f : 0.f);
  } else if (tid < 80) {
    const int v = (tid - 48) >> 4, c = c_base + (tid & 15);
    s_cf[tid] = (in_scale && c < cpad) ? (v == 0 ? in_scale[c] : in_shift[c]) : (v == 0 ? 1.f : 0.f);
  }
  for (int i = tid; i < 8 * KK; i += NT) {
    const int c = i / KK, t = i - c * KK, chn = c_base + 8 * half + c;
    s_w[i] = chn < cpad ? w[(long)t * ldw + chn] : 0.f;
  }
  float sc0 = 1.f, sc1 = 1.f, sh0 = 0.f, sh1 = 0.f;
  if (in_scale && ch < cpad) { sc0 = in_scale[ch]; sc1 = in_scale[ch + 1]; sh0 = in_shift[ch]; sh1 = in_shift[ch + 1]; }
  const bool ch0_ok = ch < g.C, ch1_ok = ch + 1 < g.C;

  CwSlots sl;
  cw_decode<P, NT, CGS>(sl, g, tid, cg);
  int sl_xo[2];   // element offset of the slot's pixel in a plane of the activated input: (image * TH + row) * xwp + 8 + column
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = sl.pp[i] >= 0 ? sl.pp[i] : 0;
    sl_xo[i] = (pp / g.W) * mg.xwp + 8 + pp % g.W;
  }

  // per-lane decode of the input-gradient tile groups, as in k_dwf_mm
  const int nl = lane & 15, q = lane >> 4;
  int t_row[MAXG], t_im[MAXG], t_pp[MAXG], t_ao[MAXG];
  unsigned t_cm[MAXG];
  {
    const int per_im = mg.nrp * mg.ncb;
#pragma unroll
    for (int G = 0; G < MAXG; ++G) {
      const int t = 16 * G + nl;
      const bool tv = t < mg.ntl;
      const int tc = tv ? t : 0;
      const int im = tc / per_im, rem = tc - im * per_im;
      const int rp = rem / mg.ncb, cb = rem - rp * mg.ncb;
      const int row = 2 * rp + (q >> 1), col = 8 * cb + 4 * (q & 1);
      t_row[G] = row;
      t_im[G] = im;
      t_pp[G] = (im * g.TH + row) * g.W + col;
      t_ao[G] = im * g.RH * g.LWp + 8 * cb + 8 * (q & 1);
      unsigned cm = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cm |= (tv && row < g.TH && col + i < g.W) ? 1u << i : 0u;
      t_cm[G] = cm;
    }
  }
  // weight-gradient operands of this lane: as A it supplies channel e = m / K shifted by b = m % K columns (m = lane & 15; rows m >= 2 K
  // repeat the last one, their results are not stored), as B window row + a of channel n / K, a = n % K
  const int mn = nl < 2 * K ? nl : 2 * K - 1;
  const int we = mn / K, wab = mn - we * K;
  // A: copy (b & 1) of channel 2 wv + e; the fragment of block (rho, kb) starts at element rho * xwp + 8 + 8 kb - b (+ 1 in copy 1)
  const bf16_t* xa_base = s_xa + ((2 * wv + we) * 2 + (wab & 1)) * mg.xplane + 8 - wab + (wab & 1);
  const bf16_t* dy_base = s_dy + (2 * wv + we) * g.plane;

  float s0a = 0.f, s0b = 0.f, s1a = 0.f, s1b = 0.f;
  f32x4 wacc = f32x4{0.f, 0.f, 0.f, 0.f};
  piece_t pfg[2], pfy[2], pfx[2];
  unsigned pfmask = 0, pxmask = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) { X::zero(pfg[i]); X::zero(pfy[i]); X::zero(pfx[i]); }

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_g = (long)slab * gss, slab_y = (long)slab * yrss, slab_x = (long)slab * xss, slab_h = (long)slab * hss;

  auto issue = [&](int n0, int hi0) {   // branch-free, see k_dwb_cw
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
  // dYraw of one piece -> the 8 channel planes
  auto put_dy = [&](const piece_t& pg_, const piece_t& py_, bool ok, bf16_t* d) {
    unsigned short* du = reinterpret_cast<unsigned short*>(d);
    float q1[8], q2[8], q3[8];
    VecIO<float, 8>::load(s_cf + cg * 8, q1);
    VecIO<float, 8>::load(s_cf + 16 + cg * 8, q2);
    VecIO<float, 8>::load(s_cf + 32 + cg * 8, q3);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const pair_t gq = X::pair(pg_, qq), yq = X::pair(py_, qq);
      const float a0 = q1[2 * qq] * X::lo(gq) + (q2[2 * qq] * X::lo(yq) + q3[2 * qq]);
      const float a1 = q1[2 * qq + 1] * X::hi(gq) + (q2[2 * qq + 1] * X::hi(yq) + q3[2 * qq + 1]);
      const pair_t pk = ok ? X::pack(a0, a1) : 0u;   // one rounding instruction and one select per channel pair
      du[(2 * qq) * g.plane] = (unsigned short)pk;
      du[(2 * qq + 1) * g.plane] = (unsigned short)(pk >> 16);
    }
  };
  auto commit = [&](int base) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0) {
        int slot = (g.ring ? 2 * P : P) + sl.rr[i] + base;
        if (slot >= g.LH) slot -= g.LH;
        put_dy(pfg[i], pfy[i], (pfmask >> i) & 1u, s_dy + sl.dyo[i] + slot * g.LWp);
        pair_t* dx_ = s_x + sl.pp[i];
        const bool okx = (pxmask >> i) & 1u;
        float q1[8], q2[8];
        VecIO<float, 8>::load(s_cf + 48 + cg * 8, q1);
        VecIO<float, 8>::load(s_cf + 64 + cg * 8, q2);
        unsigned short* xd = reinterpret_cast<unsigned short*>(s_xa + sl_xo[i]);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const pair_t xq = okx ? X::pair(pfx[i], qq) : X::zero_pair();
          dx_[qq * g.TPIXp] = xq;
          const float a0 = cw_act(X::lo(xq) * q1[2 * qq] + q2[2 * qq], in_relu, AM);
          const float a1 = cw_act(X::hi(xq) * q1[2 * qq + 1] + q2[2 * qq + 1], in_relu, AM);
          const pair_t pk = okx ? X::pack(a0, a1) : 0u;
          const unsigned short b0 = (unsigned short)pk, b1 = (unsigned short)(pk >> 16);
          xd[(4 * qq) * mg.xplane] = b0;
          xd[(4 * qq + 1) * mg.xplane + 1] = b0;
          xd[(4 * qq + 2) * mg.xplane] = b1;
          xd[(4 * qq + 3) * mg.xplane + 1] = b1;
        }
      }
    }
  };
  auto halo_sync = [&](int n0, int hi0) {
    const int npc = 2 * P * g.W;
    for (int p = tid; p < npc; p += NT) {
      const int col = p % g.W, wr = p / g.W;
      const int ho = hi0 - P + wr;
      piece_t a, b;
      X::zero(a); X::zero(b);
      const bool ok = cg_ok && ho >= 0 && ho < g.H && n0 < g.N;
      if (ok) {
        const long off = (((long)n0 * g.H + ho) * g.W + col) * 16 + cg * 8;
        X::load(a, gup + slab_g + off);
        if (yraw) X::load(b, yraw + slab_y + off);
      }
      put_dy(a, b, ok, s_dy + wr * g.LWp + col + P);
    }
  };
  auto store_h = [&](int n0, int hi0) {
    const long px = ((long)n0 * g.H + hi0) * g.W * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (sl.pp[i] >= 0 && cg_ok && n0 + sl.im[i] < g.N && hi0 + sl.rr[i] < g.H) {
        piece_t v;
        const pair_t* sx_ = s_x + sl.pp[i];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) X::set_pair(v, qq, sx_[qq * g.TPIXp]);
        X::store(v, h + slab_h + px + sl.goff[i]);
      }
    }
  };

  int tile = t_beg;
  int nb = tile / g.tiles_y, ty = tile % g.tiles_y;
  if (tile < t_end) issue(nb * g.NI, ty * g.TH);

  // Toeplitz operands of the input gradient: the forward's with the taps flipped, dx[r][c] = sum dYw[r + a][c + b] * w[K-1-a][K-1-b]
  bf16x8 ta0[NJ], ta1[NJ];
  __syncthreads();
  {
    const int orow = nl >> 3, ocol = nl & 7;
    const float* w0 = s_w + (2 * wv) * KK;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int a = 2 * j + (q >> 1) - orow;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int b = 8 * (q & 1) + e - ocol;
        const bool ok = a >= 0 && a < K && b >= 0 && b < K;
        const int idx = ok ? (K - 1 - a) * K + (K - 1 - b) : 0;
        const float v0 = w0[idx], v1 = w0[KK + idx];
        ta0[j][e] = (ok && ch0_ok) ? (bf16_t)v0 : (bf16_t)0.f;   // channels beyond C: zero operand, zero gradient
        ta1[j][e] = (ok && ch1_ok) ? (bf16_t)v1 : (bf16_t)0.f;
      }
    }
  }

  int base = 0;
  int pn0 = -1, phi0 = 0;
  const bf16_t* dy0 = s_dy + (2 * wv) * g.plane;
  const int nrows = g.NI * g.TH;
  for (; tile < t_end; ++tile) {
    const int n0 = nb * g.NI, hi0 = ty * g.TH;
    const bool fresh = g.ring && (tile == t_beg || ty == 0);
    if (fresh) base = 0;
    asm volatile("" : "+v"(sl.goff[0]), "+v"(sl.goff[1]), "+v"(sl.pp[0]), "+v"(sl.pp[1]), "+v"(sl.dyo[0]), "+v"(sl.dyo[1]));   // see k_dwb_cw
    __syncthreads();   // (A) previous tile consumed, its h complete in s_x
#pragma unroll
    for (int i = 0; i < 2; ++i) { X::touch(pfg[i]); X::touch(pfy[i]); X::touch(pfx[i]); }
    if (pn0 >= 0) store_h(pn0, phi0);
    commit(base);
    if (fresh) halo_sync(n0, hi0);
    __syncthreads();   // (B) window, activated input and pixel planes complete
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.TH);

    if (ch < cpad) {
      // ---- weight gradient: the tile's rows in groups of four (lane group q takes row 4 R + q), all 8-column blocks of the window
      // row per group: constant address steps in the inner loop, the next step's fragments in flight behind the MFMA
      if (dwp) {
        int rho = q, im = q / g.TH, r = q - im * g.TH;
        for (int R = 0; R < mg.steps; ++R) {
          const bool live = rho < nrows;
          int slot = r + wab + base;
          if (slot >= g.LH) slot -= g.LH;
          // rows past the end: the zero padding in front of row 0 (address step 0), times any finite window row
          const unsigned* ap = reinterpret_cast<const unsigned*>(live ? xa_base + rho * mg.xwp : s_xa);
          const bf16_t* bp = dy_base + (live ? (im * g.RH + slot) * g.LWp : 0);
          const int astep = live ? 4 : 0;
          u32x4 av;
          av[0] = ap[0]; av[1] = ap[1]; av[2] = ap[2]; av[3] = ap[3];
          bf16x8 bv = *reinterpret_cast<const bf16x8*>(bp);
          for (int kb = 0; kb < mg.nkb; ++kb) {
            ap += astep; bp += 8;
            u32x4 an;
            an[0] = ap[0]; an[1] = ap[1]; an[2] = ap[2]; an[3] = ap[3];   // (one block past the row on the last pass: inside the planes)
            const bf16x8 bn = *reinterpret_cast<const bf16x8*>(bp);
            wacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), bv, wacc, 0, 0, 0);
            av = an; bv = bn;
          }
          rho += 4; r += 4;
          while (r >= g.TH) { r -= g.TH; ++im; }
        }
      }
      // ---- input gradient
#pragma unroll
      for (int G = 0; G < MAXG; ++G) {
        if (G < mg.ngroups) {
          f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            int rr_ = t_row[G] + 2 * j + base;
            if (rr_ >= g.LH) rr_ -= g.LH;
            const bf16_t* bp = dy0 + t_ao[G] + rr_ * g.LWp;
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bp);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bp + g.plane);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta0[j], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta1[j], b1, acc1, 0, 0, 0);
          }
          const bool ok = n0 + t_im[G] < g.N && hi0 + t_row[G] < g.H;
          const unsigned cm = ok ? t_cm[G] : 0u;
          pair_t* xp = s_x + wv * g.TPIXp + t_pp[G];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if ((cm >> i) & 1u) {
              // epilogue: activation backward of the producer, rounding, statistics; h replaces x in its LDS slot
              const pair_t xq = xp[i];
              const float x0 = X::lo(xq), x1 = X::hi(xq);
              float v0 = cw_act_bwd(acc0[i], x0 * sc0 + sh0, in_relu, AM);
              float v1 = cw_act_bwd(acc1[i], x1 * sc1 + sh1, in_relu, AM);
              const pair_t o = X::pack(v0, v1);
              v0 = X::lo(o); v1 = X::hi(o);   // statistics of the stored (rounded) values
              s0a += v0; s0b += v1;
              s1a += v0 * x0; s1b += v1 * x1;
              xp[i] = o;
            }
          }
        }
      }
    }
    pn0 = n0; phi0 = hi0;
    nb = nnb; ty = nty;
    if (g.ring) { base += g.TH; if (base >= g.LH) base -= g.LH; }
  }
  __syncthreads();
  if (pn0 >= 0) store_h(pn0, phi0);

  // weight-gradient partial row of this worker: lane (n, q) holds D[m = 4 q + i][n]; the (e, b) x (e, a) entries are
  // dW[channel ch + e][K-1-a][K-1-b]
  if (dwp && nl < 2 * K) {
    const int e = nl / K, a = nl - e * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * q + i;
      const int em = m / K, b = m - em * K;
      if (m < 2 * K && em == e && ch + e < g.C)
        dwp[((long)worker * g.C + ch + e) * KK + (K - 1 - a) * K + (K - 1 - b)] = wacc[i];
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

// ---------------------------------------------------------------------------------------------------------------- host side
// The pixel tiles, workers and staging slots are cw_geometry's (so W is a multiple of 7 here as well); the window becomes per-CHANNEL
// planes of 16-bit elements [image][row][column] with a row pitch of LWp elements: 8 columns per MFMA column block + 8 (the patch of the
// last block reaches 16 columns), in 16-byte units an odd number so that the patch rows of neighbouring tiles start in different bank
// groups.  Backward: additionally two copies of the activated input, zero-padded on the left, [image * TH + row][xwp] per channel.
static void mm_derive(CwGeom& g, MmGeom& mg, int W, int K) {
  mg.nrp = (g.TH + 1) / 2;
  mg.ncb = (W + 7) / 8;
  mg.ntl = g.NI * mg.nrp * mg.ncb;
  mg.ngroups = (mg.ntl + 15) / 16;
  g.LH = 2 * mg.nrp + K - 1;
  g.RH = g.LH;
  int u = mg.ncb + 1;
  if (mg.ncb > 1 && (u & 1) == 0) ++u;
  g.LWp = 8 * u;
  g.plane = g.NI * g.RH * g.LWp;   // elements per channel plane (a multiple of 8: 16-byte aligned planes)
  g.TPIX = g.NI * g.TH * W;
  int tp = g.TPIX;
  while (tp % 8 != 4) ++tp;
  g.TPIXp = tp;
  g.ntiles = ((g.N + g.NI - 1) / g.NI) * g.tiles_y;
  mg.nkb = (W + K - 1 + 7) / 8;
  mg.xwp = 8 * (mg.nkb + 1);
  mg.xplane = g.NI * g.TH * mg.xwp;
  mg.steps = (g.NI * g.TH + 3) / 4;
}
static size_t mm_lds(const CwGeom& g, const MmGeom& mg, int K, bool bwd) {
  size_t b = (size_t)8 * g.plane * 2 + (size_t)4 * g.TPIXp * sizeof(unsigned) + (32 + 8 * K * K) * sizeof(float);
  if (bwd) b += (size_t)8 * 2 * mg.xplane * 2 + 48 * sizeof(float);
  return b;
}
static bool mm_geometry(CwGeom& g, MmGeom& mg, int N, int H, int W, int C, int K, bool bwd) {
  if (!cw_geometry(g, N, H, W, C, K)) return false;
  if (g.ring && (g.TH & 1)) return false;   // row pairs must not straddle the ring's tile boundary
  mm_derive(g, mg, W, K);
  // small maps (whole images per tile): fewer images per tile where the backward's planes would leave one workgroup per CU
  while (bwd && !g.ring && g.NI > 1 && mm_lds(g, mg, K, true) > (size_t)52 * 1024) {
    --g.NI;
    mm_derive(g, mg, W, K);
  }
  return mg.ngroups <= MM_MAXG;
}

static int mm_mode() {
  // bits 0-2: forward k = 3 / 5 / 7, bits 3-5: backward k = 3 / 5 / 7, bit 6: backward also on whole-image tiles (14 x 14, 7 x 7 maps).
  // Default (profiles/r05_dw_mm_per_shape.txt, batch 256): forward k = 5, 7 everywhere (k = 3 is as fast on the packed-FMA rows);
  // backward k = 7 on row-ring tiles (56 x 56: 0.42 -> 0.34 ms, 28 x 28: 0.20 -> 0.16 ms) -- a tile of the backward kernel costs about
  // the same for every k (commit + two operand copies + epilogue, ~1000 instructions per wave), which beats the packed-FMA rows only
  // at k = 7 and only where a worker walks many tiles.
  static const int m = getenv("ATOMNAS_DW_MM") ? atoi(getenv("ATOMNAS_DW_MM")) : 38;
  return m;
}
static bool mm_bwd_wanted(const CwGeom& g, int k) {
  const int bit = k == 3 ? 8 : (k == 5 ? 16 : 32);
  return (mm_mode() & bit) && (g.ring || (mm_mode() & 64));
}

template <int K>
static int mm_launch_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                         float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  MmGeom mg;
  if (!mm_geometry(g, mg, N, H, W, C, K, false)) return -1;
  const size_t lds = mm_lds(g, mg, K, false);
  if (lds > max_lds_bytes()) return -1;
#define MM_FWD(AMV)                                                                                                         \
  {                                                                                                                         \
    auto kern = k_dwf_mm<K, AMV, 4>;                                                                                        \
    cw_workers(g, resident_per_cu(kern, 256, lds), stats ? stat_rows : 0, 4);                                               \
    hipLaunchKernelGGL(kern, dim3(cw_grid(g, 4)), dim3(256), lds, st, (const bf16_t*)x, xss, sc, sh, relu, w, ldw, (bf16_t*)y, yss, \
                       stats, stat_ld, stat_rows, g, mg);                                                                   \
  }
  if (relu == ACT_RELU6) MM_FWD(ACT_RELU6) else if (relu == ACT_SWISH) MM_FWD(ACT_SWISH) else if (relu == ACT_RELU && sc) MM_FWD(ACT_RELU) else MM_FWD(0)
#undef MM_FWD
  return check_launch("dwconv_fwd(mm)");
}

// -1: not one of this file's cases (the caller continues with dwconv_cw.hip / dwconv.hip); otherwise the launch status
int dwconv_mm_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                  float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k, int dtype, hipStream_t st) {
  if (dtype != DT_BF16 || xss == 0 || yss == 0 || ldw < ((C + 7) & ~7)) return -1;
  const int bit = k == 3 ? 1 : (k == 5 ? 2 : 4);
  if (!(mm_mode() & bit)) return -1;
#define MM_F(KV) return mm_launch_fwd<KV>(x, xss, sc, sh, relu, w, ldw, y, yss, stats, stat_ld, stat_rows, N, H, W, C, st)
  if (k == 3) MM_F(3);
  if (k == 5) MM_F(5);
  if (k == 7) MM_F(7);
#undef MM_F
  return -1;
}

template <int K>
static int mm_launch_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                         const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                         float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  MmGeom mg;
  if (!mm_geometry(g, mg, N, H, W, C, K, true) || !mm_bwd_wanted(g, K)) return -1;
  const size_t lds = mm_lds(g, mg, K, true);
  if (lds > max_lds_bytes()) return -1;
  const bool two = mg.ngroups <= 2;   // row-ring tiles have at most two MFMA tile groups: the instance with two per-lane decodes (10 registers less: no spill at k = 7)
#define MM_BWD(AMV)                                                                                                         \
  {                                                                                                                         \
    auto kern = two ? k_dwb_mm<K, AMV, 3, 2> : k_dwb_mm<K, AMV, 3, MM_MAXG>;                                                                                      \
    cw_workers(g, resident_per_cu(kern, 256, lds), (stats || dw) ? part_rows : 0, 4);                                       \
    hipLaunchKernelGGL(kern, dim3(cw_grid(g, 4)), dim3(256), lds, st, (const bf16_t*)gup, gss, (const bf16_t*)yraw, yrss, c1, c2, c3, \
                       (const bf16_t*)x, xss, sc, sh, relu, w, ldw, (bf16_t*)h, hss, dw ? dw_ws : nullptr, stats, stat_ld, part_rows, g, mg); \
  }
  if (relu == ACT_RELU6) MM_BWD(ACT_RELU6) else if (relu == ACT_SWISH) MM_BWD(ACT_SWISH) else if (relu == ACT_RELU && sc) MM_BWD(ACT_RELU) else MM_BWD(0)
#undef MM_BWD
  if (int rc = check_launch("dwconv_bwd(mm)")) return rc;
  if (dw) return reduce_parts(dw_ws, (long)C * K * K, g.nworkers, (long)C * K * K, dw, C * K * K, 0, 1, st);
  return 0;
}

int dwconv_mm_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                  const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                  float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k, int stride,
                  int dtype, hipStream_t st) {
  if (dtype != DT_BF16 || stride != 1 || gss == 0 || xss == 0 || hss == 0 || (yraw && yrss == 0) || ldw < ((C + 7) & ~7)) return -1;
#define MM_B(KV) return mm_launch_bwd<KV>(gup, gss, yraw, yrss, c1, c2, c3, x, xss, sc, sh, relu, w, ldw, h, hss, dw, stats, stat_ld, part_rows, dw_ws, N, H, W, C, st)
  if (k == 3) MM_B(3);
  if (k == 5) MM_B(5);
  if (k == 7) MM_B(7);
#undef MM_B
  return -1;
}

int dwconv_mm_supported(int N, int H, int W, int C, int k, int dir) {
  CwGeom g;
  MmGeom mg;
  if (!(k == 3 || k == 5 || k == 7)) return 0;
  const int bit = k == 3 ? 1 : (k == 5 ? 2 : 4);
  if (!mm_geometry(g, mg, N, H, W, C, k, dir != 0)) return 0;
  if (dir ? !mm_bwd_wanted(g, k) : !(mm_mode() & bit)) return 0;
  return mm_lds(g, mg, k, dir != 0) <= max_lds_bytes() ? 1 : 0;
}

}  // namespace atomnas

// 1 when atomnas_dwconv_fwd (dir = 0) / atomnas_dwconv_bwd (dir = 1) take the matrix-core kernels of this file for the shape (bf16
// slab-major tensors; stride 2: the forward of dwconv_mm2.hip).  Tests build the oracle's storage model from it (oracle/atomnas_oracle.py bf16_storage_mm).
extern "C" int atomnas_dwconv_mm_supported(int N, int H, int W, int C, int k, int stride, int dtype, int dir) {
  if (dtype != atomnas::DT_BF16) return 0;
  if (stride == 2) return dir == 0 ? atomnas::dwconv_mm2_supported(N, H, W, C, k) : 0;
  return stride == 1 ? atomnas::dwconv_mm_supported(N, H, W, C, k, dir) : 0;
}

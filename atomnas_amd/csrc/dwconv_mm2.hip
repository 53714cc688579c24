// Depthwise k x k convolution, STRIDE 2, forward, with the tap arithmetic on the matrix cores (gfx950), bf16 storage, slab-major tensors.
//
// The stride-2 depthwise layers (models/mobilenet_base.py:330-336 with stride 2: the first block of a stage) read four input pixels per
// output pixel: 0.8 of their bytes are the INPUT, and the tile kernel of dwconv.hip that served them (square tiles with halo re-reads,
// per-tap scalar FMAs) moved 2.4 - 3.4 TB/s (profiles/r05_bs256_per_shape_timing.txt).  This is dwconv_mm.hip's forward kernel on
// the stride-2 geometry:
//
//     one MFMA "tile" = 2 output rows x 8 output columns of one channel;  its input patch is (k + 2) rows x 24 columns of the activated
//     input window (2 * 7 + k <= 21 columns are used), cut into k-blocks of 8 consecutive columns of one row: 3 (k + 2) k-blocks, four
//     per v_mfma_f32_16x16x32_f16, k-block kb = patch row kb / 3, column block kb % 3:
//         D[m = (orow, ocol)][n = tile] += sum_k A[m][k = (prow, pcol)] * B[k][n],   A[m][k] = w[prow - 2 orow][pcol - 2 ocol] or 0
//     k = 7: 7 MFMAs per 16 tiles and channel, k = 5: 6, k = 3: 4.  Patches start at multiples of 16 window columns: 16-byte aligned
//     reads (neighbouring tiles are 32 bytes apart: a two-way bank conflict on the B reads, 28 reads per wave and pixel tile -- nothing
//     next to the 56 staging writes).
//
// Tiles are full-width bands of THd output rows = 2 THd input rows; the window rows are a ring, a band re-uses the k - 2 rows it shares
// with the band above (no halo re-reads from HBM); small maps take whole images per tile.  A thread stages 7 input pieces (8 channels
// of a pixel, 16 bytes) per tile, prefetched into registers during the previous tile's arithmetic, and stores up to 2 output pieces.
//
// Numerics are dwconv_mm.hip's forward: fp16 operands (activated input, taps), exact products, fp32 accumulation, bf16 output;
// oracle/atomnas_oracle.py restates the roundings (bf16_storage_mm).  Bit-reproducible; no atomics.
#include "dwconv_cw.h"

namespace atomnas {

typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int M2_MAXG = 2;   // MFMA tile groups (16 tiles each) of a pixel tile
constexpr int M2_XS = 7;     // staging slots (input pieces) per thread and tile: 4 x the (<= 448) output pixels / 256 threads

struct Mm2Geom {
  int nrp, ncb, ntl, ngroups;   // row pairs / 8-column blocks of the OUTPUT band per image, MFMA tiles per pixel tile, groups of 16
  int TPIXDp;                   // pitch of the output pixel planes
};

__device__ __forceinline__ float mm2_clamp16(float a) { return __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); }

template <int K, int AM, int WPS, int NG>
__global__ __launch_bounds__(256, WPS) void k_dwf_mm2(const bf16_t* __restrict__ x, long xss, const float* __restrict__ in_scale,
                                                   const float* __restrict__ in_shift, int in_relu, const float* __restrict__ w, int ldw,
                                                   bf16_t* __restrict__ y, long yss, float* __restrict__ stats, int stat_ld, int stat_rows,
                                                   CwGeom g, Mm2Geom mg) {
  typedef Cw<bf16_t> X;
  typedef X::pair_t pair_t;
  typedef X::piece_t piece_t;
  constexpr int P = (K - 1) / 2, NW = 4, NT = 256, MAXG = NG;   // NG = MFMA tile groups of the pixel tile (1 or 2)
  constexpr int NKB = 3 * (K + 2), NJ = (NKB + 3) / 4;   // k-blocks of a patch, MFMAs per tile
  constexpr int HALO = K - 2;                            // input rows a band shares with the band above
  constexpr int HREAL = HALO - P;                        // ... of which rows 0 .. P - 2 of the image are real for the image's first band
  constexpr int TPW = 40, TPC = (K + 5) * TPW;          // zero-padded tap table of a channel: rows ky = -2 .. K + 2, columns kx = -14 .. 25
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f16_t* s_in = reinterpret_cast<f16_t*>(smem);                          // [8 channels][plane]: activated input window, fp16
  pair_t* s_y = reinterpret_cast<pair_t*>(s_in + 8 * g.plane);            // [4 pairs][TPIXDp]: the tile's output
  float* s_cf = reinterpret_cast<float*>(s_y + NW * mg.TPIXDp);           // [2][16] scale / shift of the slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int slab, worker, half;
  if (!cw_block<NW>(g, slab, worker, half)) return;
  const int c_base = slab * 16;
  const int ch = c_base + 2 * (wv + 4 * half);
  const int cpad = (g.C + 7) & ~7;
  const int cg = half;
  const bool cg_ok = c_base + cg * 8 < cpad;

  if (tid < 32) {
    const int v = tid >> 4, c = c_base + (tid & 15);
    s_cf[tid] = (in_scale && c < cpad) ? (v == 0 ? in_scale[c] : in_shift[c]) : (v == 0 ? 1.f : 0.f);
  }
  // zero-padded tap table of the workgroup's 8 channels (in the window planes, which are not in use yet): row ky + 2, column kx + 14
  float* s_tp = reinterpret_cast<float*>(s_in);
  {
    f32x4* z = reinterpret_cast<f32x4*>(s_tp);
    for (int i = tid; i < 2 * TPC; i += NT) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  for (int i = tid; i < 8 * K * K; i += NT) {
    const int c = i / (K * K), t = i - c * (K * K), chn = c_base + 8 * half + c;
    if (chn < g.C) s_tp[c * TPC + (t / K + 2) * TPW + t % K + 14] = w[(long)t * ldw + chn];   // channels beyond C: zero operand, zero output
  }

  // Tile-independent decode of the staging slots.  A tile's input pieces (8 channels of a pixel) run over (image, input row, column) and
  // are contiguous in the slab: piece tid + 256 i is at 16 (tid + 256 i) elements from the tile's first one, and the VALID pieces
  // (rows inside the image, images inside the batch) are a prefix -- one comparison per slot.  Only the window offset needs the decode.
  int x_wo[M2_XS];   // im * RH * LWp + row * LWp + col + P
#pragma unroll
  for (int i = 0; i < M2_XS; ++i) {
    const int pp = tid + i * NT;
    const int col = pp % g.W, t2 = pp / g.W;
    const int rr = t2 % g.TH, im = t2 / g.TH;
    x_wo[i] = (im * g.RH + rr) * g.LWp + col + P;
  }

  // per-lane decode of the MFMA tile groups: lane (n = lane & 15, q = lane >> 4) of group G works on tile 16 G + n; as the B operand of
  // MFMA j it supplies k-block 4 j + q of the tile's patch, as the D operand it receives output row q >> 1, columns 4 (q & 1) .. + 3
  const int nl = lane & 15, q = lane >> 4;
  int t_r4[MAXG], t_ri[MAXG], t_pp[MAXG], t_ao[MAXG];
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
      t_r4[G] = 4 * rp;
      t_ri[G] = row | (im << 16);
      t_pp[G] = (im * g.THd + row) * g.Wo + col;
      t_ao[G] = im * g.RH * g.LWp + 16 * cb;
      unsigned cm = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cm |= (tv && row < g.THd && col + i < g.Wo) ? 1u << i : 0u;
      t_cm[G] = cm;
    }
  }
  // k-block of MFMA j for this lane: patch row and column offset (k-blocks beyond the patch: the operand A is zero, read the last one)
  int kb_row[NJ], kb_col[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int kb = (4 * j + q < NKB) ? 4 * j + q : NKB - 1;
    kb_row[j] = kb / 3;
    kb_col[j] = 8 * (kb - 3 * (kb / 3));
  }

  const int h_wo = (P + tid / g.W) * g.LWp + tid % g.W + P;   // window offset of piece tid of the image's first rows (first band's halo)

  float sa = 0.f, sb = 0.f, qa = 0.f, qb = 0.f;
  piece_t pfx[M2_XS], pfh;
  int pxn = 0, phn = 0;   // valid pieces of the prefetched tile / of its prefetched halo rows
  X::zero(pfh);
#pragma unroll
  for (int i = 0; i < M2_XS; ++i) X::zero(pfx[i]);

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  const long slab_x = (long)slab * xss, slab_y = (long)slab * yss;

  // tile (n0, hd0): images n0 .., output rows hd0 ..; ring: its NEW input rows start at 2 hd0 - P + HALO
  auto issue = [&](int n0, int hd0) {   // branch-free: invalid pieces load the tile's first piece
    const int hi_s = g.ring ? 2 * hd0 - P + HALO : 0;
    const int rows_ok = g.H - hi_s < g.TH ? g.H - hi_s : g.TH, ims_ok = g.N - n0 < g.NI ? g.N - n0 : g.NI;
    pxn = cg_ok ? (g.ring ? rows_ok * g.W : ims_ok * g.TH * g.W) : 0;
    const bf16_t* src = x + slab_x + (pxn > 0 ? ((long)n0 * g.H + hi_s) * g.W * 16 : 0) + cg * 8;   // (no valid piece: the slab's first one)
#pragma unroll
    for (int i = 0; i < M2_XS; ++i) {
      const int pp = tid + i * NT;
      X::load(pfx[i], src + (pp < pxn ? (unsigned)pp * 16u : 0u));
    }
    // the first band of an image: rows 0 .. P - 2 are the real rows of its halo (the other P are above the image) -- prefetched with
    // the tile, so that the start of an image costs no exposed load latency (28 x 28 output maps: every other tile starts an image)
    if (HREAL > 0 && g.ring && hd0 == 0) {
      phn = cg_ok ? HREAL * g.W : 0;
      X::load(pfh, x + slab_x + (long)n0 * g.H * g.W * 16 + cg * 8 + (tid < phn ? (unsigned)tid * 16u : 0u));
    }
  };
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
    // window row of tile row rr: (ring ? HALO : P) + rr + base, modulo LH (ring: rr = pp / W, the rows that wrap are a suffix of the pieces)
    const int rowbase = (g.ring ? HALO : P) + base;
    const int wrap_pp = g.ring ? (g.LH - rowbase) * g.W : 0x7fffffff;
    const int o0 = rowbase * g.LWp, o1 = o0 - g.LH * g.LWp;
#pragma unroll
    for (int i = 0; i < M2_XS; ++i) {
      const int pp = tid + i * NT;
      if (pp < g.TPIX) put_in(pfx[i], pp < pxn, s_in + x_wo[i] + (pp >= wrap_pp ? o1 : o0));
    }
  };
  auto halo_top = [&]() {   // first band of an image: P zero rows, then the prefetched rows 0 .. P - 2
    for (int wr = 0; wr < P; ++wr) {
      for (int col = tid; col < g.W; col += NT) {
        f16_t* d = s_in + wr * g.LWp + col + P;
#pragma unroll
        for (int c = 0; c < 8; ++c) d[c * g.plane] = (f16_t)0.f;
      }
    }
    if (HREAL > 0 && tid < HREAL * g.W) put_in(pfh, tid < phn, s_in + h_wo);
  };
  auto store_y = [&](int n0, int hd0) {   // the tile's output pieces are contiguous in the slab as well
    const int rows_ok = g.Ho - hd0 < g.THd ? g.Ho - hd0 : g.THd, ims_ok = g.N - n0 < g.NI ? g.N - n0 : g.NI;
    const int nv = cg_ok ? (g.ring ? rows_ok * g.Wo : ims_ok * g.THd * g.Wo) : 0;
    bf16_t* dst = y + slab_y + ((long)n0 * g.Ho + hd0) * g.Wo * 16 + cg * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = tid + i * NT;
      if (pp < nv) {
        piece_t v;
        const pair_t* sy_ = s_y + pp;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) X::set_pair(v, qq, sy_[qq * mg.TPIXDp]);
        X::store(v, dst + (unsigned)pp * 16u);
      }
    }
  };

  int tile = t_beg;
  int nb = tile / g.tiles_y, ty = tile % g.tiles_y;
  // a worker that starts inside an image: the HALO rows above its first band's new rows, loaded now, written after the planes are zeroed
  constexpr int HS = 3;   // HALO * W <= 3 * 256 pieces (host side)
  piece_t pfs[HS];
  const bool start_mid = g.ring && ty != 0 && tile < t_end;
  if (start_mid) {
#pragma unroll
    for (int i = 0; i < HS; ++i) {
      const int pp = tid + i * NT;
      const bool ok = cg_ok && pp < HALO * g.W;
      X::load(pfs[i], x + slab_x + ((long)(nb * g.NI) * g.H + 2 * ty * g.THd - P) * g.W * 16 + cg * 8 + (ok ? (unsigned)pp * 16u : 0u));
    }
  }
  if (tile < t_end) issue(nb * g.NI, ty * g.THd);

  // Toeplitz operands of the wave's two channels: lane (m = lane & 15, q) holds k-block 4 j + q of MFMA j (k-blocks beyond the patch
  // fall into the table's zero rows)
  f16x8 ta0[NJ], ta1[NJ];
  __syncthreads();
  {
    const int orow = nl >> 3, ocol = nl & 7;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int kb = 4 * j + q;
      const float* t0 = s_tp + (2 * wv) * TPC + (kb / 3 - 2 * orow + 2) * TPW + 8 * (kb % 3) - 2 * ocol + 14;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ta0[j][e] = (f16_t)t0[e];
        ta1[j][e] = (f16_t)t0[TPC + e];
      }
    }
  }
  __syncthreads();
  {   // halo columns / rows outside the image stay zero
    u32x4* z = reinterpret_cast<u32x4*>(s_in);
    for (int i = tid; i < g.plane; i += NT) z[i] = u32x4{0u, 0u, 0u, 0u};
  }
  if (start_mid) {   // (2 ty THd - P >= 1: the rows exist)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < HS; ++i) {
      const int pp = tid + i * NT;
      if (pp < HALO * g.W) put_in(pfs[i], cg_ok, s_in + (pp / g.W) * g.LWp + pp % g.W + P);
    }
  }

  int base = 0;
  int pn0 = -1, phd0 = 0;
  const f16_t* in0 = s_in + (2 * wv) * g.plane;
  for (; tile < t_end; ++tile) {
    const int n0 = nb * g.NI, hd0 = ty * g.THd;
    const bool fresh = g.ring && (tile == t_beg || ty == 0);
    if (fresh) base = 0;
    __syncthreads();   // (A) previous tile consumed, its output complete in s_y
#pragma unroll
    for (int i = 0; i < M2_XS; ++i) X::touch(pfx[i]);
    X::touch(pfh);
    if (pn0 >= 0) store_y(pn0, phd0);
    commit(base);
    if (fresh && ty == 0) halo_top();
    __syncthreads();   // (B)
    int nnb = nb, nty = ty + 1;
    if (nty == g.tiles_y) { nty = 0; ++nnb; }
    if (tile + 1 < t_end) issue(nnb * g.NI, nty * g.THd);

    if (ch < cpad) {
      // both tile groups side by side (independent accumulator chains), the B fragments of MFMA j + 1 read while MFMA j runs
      f32x4 acc0[NG], acc1[NG];
      f16x8 bq[2][NG][2];
      auto read_b = [&](int j, int slot) {
#pragma unroll
        for (int G = 0; G < NG; ++G) {
          int r = t_r4[G] + kb_row[j] + base;
          if (r >= g.LH) r -= g.LH;
          const f16_t* bp = in0 + t_ao[G] + kb_col[j] + r * g.LWp;
          bq[slot][G][0] = *reinterpret_cast<const f16x8*>(bp);
          bq[slot][G][1] = *reinterpret_cast<const f16x8*>(bp + g.plane);
        }
      };
#pragma unroll
      for (int G = 0; G < NG; ++G) acc0[G] = acc1[G] = f32x4{0.f, 0.f, 0.f, 0.f};
      read_b(0, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j + 1 < NJ) read_b(j + 1, (j + 1) & 1);
#pragma unroll
        for (int G = 0; G < NG; ++G) {
          acc0[G] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta0[j], bq[j & 1][G][0], acc0[G], 0, 0, 0);
          acc1[G] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta1[j], bq[j & 1][G][1], acc1[G], 0, 0, 0);
        }
      }
#pragma unroll
      for (int G = 0; G < NG; ++G) {
        const bool ok = n0 + (t_ri[G] >> 16) < g.N && hd0 + (t_ri[G] & 0xffff) < g.Ho;
        const unsigned cm = ok ? t_cm[G] : 0u;
        pair_t* yp = s_y + wv * mg.TPIXDp + t_pp[G];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if ((cm >> i) & 1u) {
            const pair_t o = X::pack(acc0[G][i], acc1[G][i]);
            const float v0 = X::lo(o), v1 = X::hi(o);
            sa += v0; sb += v1; qa += v0 * v0; qb += v1 * v1;
            yp[i] = o;
          }
        }
      }
    }
    pn0 = n0; phd0 = hd0;
    nb = nnb; ty = nty;
    if (g.ring) { base += g.TH; if (base >= g.LH) base -= g.LH; }
  }
  __syncthreads();
  if (pn0 >= 0) store_y(pn0, phd0);

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

// ---------------------------------------------------------------------------------------------------------------- host side
// Window planes of fp16 elements [image][row][column]: LH = 4 nrp + k - 2 rows (ring: exactly the 2 THd new rows + the k - 2 shared ones),
// row pitch 16 ncb + 8 elements (patches of 24 columns every 16; in 16-byte units an odd number).
static void mm2_derive(CwGeom& g, Mm2Geom& mg, int K) {
  mg.nrp = (g.THd + 1) / 2;
  mg.ncb = (g.Wo + 7) / 8;
  mg.ntl = g.NI * mg.nrp * mg.ncb;
  mg.ngroups = (mg.ntl + 15) / 16;
  g.TH = 2 * g.THd;
  g.LH = 4 * mg.nrp + K - 2;
  g.RH = g.LH;
  g.LWp = 16 * mg.ncb + 8;
  g.plane = g.NI * g.RH * g.LWp;
  if (g.plane < (K + 5) * 80) g.plane = (K + 5) * 80;   // the prologue's tap table (8 channels x (K + 5) x 40 floats) lives in the planes
  g.TPIX = g.NI * g.TH * g.W;
  g.TPIXD = g.NI * g.THd * g.Wo;
  int tp = g.TPIXD;
  while (tp % 8 != 4) ++tp;
  mg.TPIXDp = tp;
  g.TPIXp = tp;
  g.ntiles = ((g.N + g.NI - 1) / g.NI) * g.tiles_y;
}
static size_t mm2_lds(const CwGeom& g, const Mm2Geom& mg, int K) {
  return (size_t)8 * g.plane * 2 + (size_t)4 * mg.TPIXDp * sizeof(unsigned) + 32 * sizeof(float);
}
static bool mm2_geometry(CwGeom& g, Mm2Geom& mg, int N, int H, int W, int C, int K) {
  if (!cw2_geometry(g, N, H, W, C, K)) return false;
  if (g.ring && (g.THd & 1)) return false;   // row pairs must not straddle the ring's band boundary
  mm2_derive(g, mg, K);
  // small maps (whole images per tile): fewer images per tile where the planes would leave two workgroups per CU
  while (!g.ring && g.NI > 1 && (mm2_lds(g, mg, K) > (size_t)52 * 1024 || mg.ngroups > M2_MAXG)) {
    --g.NI;
    mm2_derive(g, mg, K);
  }
  if (mg.ngroups > M2_MAXG || g.TPIX > M2_XS * 256 || g.TPIXD > 2 * 256) return false;
  if (g.ring && ((K - 2 - (K - 1) / 2) * W > 256 || (K - 2) * W > 3 * 256)) return false;   // prefetch slots of the halo rows
  // k = 3 has too little arithmetic to win on the short tile walks of the small maps (profiles/r05_dw_mm2_per_shape.txt: 28 x 28 input
  // 0.037 -> 0.040 ms, 14 x 14 0.028 -> 0.040 ms where k = 5 / 7 gain 10 - 25 %): the tile kernel of dwconv.hip keeps those
  return K > 3 || g.tiles_y > 2;
}

static int mm2_mode() {
  // bit 7 of ATOMNAS_DW_MM (see dwconv_mm.hip): the stride-2 forward on the matrix cores
  static const int m = getenv("ATOMNAS_DW_MM") ? atoi(getenv("ATOMNAS_DW_MM")) : 128;
  return m & 128;
}

template <int K>
static int mm2_launch_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                          float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, hipStream_t st) {
  CwGeom g;
  Mm2Geom mg;
  if (!mm2_geometry(g, mg, N, H, W, C, K)) return -1;
  const size_t lds = mm2_lds(g, mg, K);
  if (lds > max_lds_bytes()) return -1;
#define MM2_FWD(AMV)                                                                                                        \
  {                                                                                                                         \
    auto kern = mg.ngroups == 1 ? k_dwf_mm2<K, AMV, 2, 1> : k_dwf_mm2<K, AMV, 2, 2>;                                        \
    cw_workers(g, resident_per_cu(kern, 256, lds), stats ? stat_rows : 0, 4);                                               \
    hipLaunchKernelGGL(kern, dim3(cw_grid(g, 4)), dim3(256), lds, st, (const bf16_t*)x, xss, sc, sh, relu, w, ldw, (bf16_t*)y, yss, \
                       stats, stat_ld, stat_rows, g, mg);                                                                   \
  }
  if (relu == ACT_RELU6) MM2_FWD(ACT_RELU6) else if (relu == ACT_SWISH) MM2_FWD(ACT_SWISH) else if (relu == ACT_RELU && sc) MM2_FWD(ACT_RELU) else MM2_FWD(0)
#undef MM2_FWD
  return check_launch("dwconv_fwd(mm2)");
}

// -1: not one of this file's cases (the caller continues with the tile kernels of dwconv.hip); otherwise the launch status
int dwconv_mm2_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                   float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k, int dtype, hipStream_t st) {
  if (dtype != DT_BF16 || xss == 0 || yss == 0 || ldw < ((C + 7) & ~7) || !mm2_mode()) return -1;
#define MM2_F(KV) return mm2_launch_fwd<KV>(x, xss, sc, sh, relu, w, ldw, y, yss, stats, stat_ld, stat_rows, N, H, W, C, st)
  if (k == 3) MM2_F(3);
  if (k == 5) MM2_F(5);
  if (k == 7) MM2_F(7);
#undef MM2_F
  return -1;
}

int dwconv_mm2_supported(int N, int H, int W, int C, int k) {
  CwGeom g;
  Mm2Geom mg;
  if (!(k == 3 || k == 5 || k == 7) || !mm2_mode()) return 0;
  if (!mm2_geometry(g, mg, N, H, W, C, k)) return 0;
  return mm2_lds(g, mg, k) <= max_lds_bytes() ? 1 : 0;
}

}  // namespace atomnas

// Expand 1x1 + BatchNorm + activation computed ON CHIP in front of the depthwise convolution ("E-elimination", gfx950, bf16).
//
// The atomic block (models/mobilenet_base.py:316-336, 371-382) expands its narrow input x [M][inp] to the 6x wider E = x W1^T,
// normalises and activates it, and runs a depthwise k x k convolution over the result.  With the stand-alone kernels E makes
// four trips through HBM per training step (written by the expand GEMM, read by the depthwise forward, read by the depthwise
// backward, read by the expand backward) although it is a 16 x 16 x 32 MFMA away from x.  The kernels of this file never see
// E in HBM:
//   * k_xdwf (forward):  a workgroup owns a tile of pixels and a column of S channel slabs.  Per slab it computes the activated
//     expand output of the tile's window (tile rows + k - 1 halo rows) with mfma_f32_16x16x32_bf16 -- weights as the A operand,
//     pixels of x as the B operand straight from global memory (x is 6x narrower than E and stays in L2 across the slabs) --
//     applies scale / shift / activation in the accumulator registers and writes the fp32 channel-pair planes the tap loop of
//     dwconv_cw.hip reads.  The tap loop, the output staging and the statistics epilogue are those of k_dwf_cw.
//   * the statistics of the expand BatchNorm come from the inp x inp Gram matrix of x (sum e_c = w_c . sum x,
//     sum e_c^2 = w_c^T (X^T X) w_c): k_gram_stats; no pass over a hidden-size tensor.
//   * k_xdwb (backward): k_dwb_cw with the raw expand output of the tile recomputed the same way instead of being read.
//   * k_xb_coeffs: the expand backward without E.  With dE = c1 h + c2 E + c3 (BatchNorm backward),
//         dX  = (c1 h) W1 + x M + v,          M = W1^T diag(c2) W1,  v = c3^T W1
//         dW1 = diag(c1) h^T x + diag(c2) W1 (X^T X) + c3 (sum x)^T
//     so one pass over h (existing GEMM kernels, c1 as their scale prologue) plus inp x inp sized corrections.
// Numerics: E is never rounded to the storage type (it used to be stored as bf16); everything else as in dwconv_cw.hip.
// No atomics; all reductions in a fixed order.
#include "../common.h"
#include "xdw_internal.h"
#include <cstdio>
#include <cstdlib>

#ifndef XD_KO
#define XD_KO 0   // experiment builds: knock-outs of k_xdwf (1: no output stores, 2: no tap loop, 4: no window writes, 8: no pixel loads, 16: no barriers' work between)
#endif
#ifndef XD_TIMING
#define XD_TIMING 0   // experiment builds (tools/variant.sh): cycle accounting of the phases of a slab-tile in k_xdwf
#endif

namespace atomnas {

#if XD_TIMING
__device__ unsigned long long g_xd_timing[8];
#define XDMARK(i)                                                    \
  {                                                                  \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long tn_ = __builtin_readcyclecounter();     \
    tacc[i] += tn_ - tlast;                                          \
    tlast = tn_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                               \
  }
#else
#define XDMARK(i)
#endif

typedef __attribute__((ext_vector_type(4))) unsigned xu32x4;

struct XdGeom {
  int N, H, W, C, inp;
  int TH, NI, tiles_y, ns;     // tile: NI images x TH rows x W columns; ns = W / 7 strips per row
  int LH, LWp, plane;          // window rows per image (TH + K - 1), row pitch, elements (f32x2) per channel-pair plane
  int TPIX, TPIXp;             // output pixels per tile; pitch of the output planes
  int NG;                      // 16-pixel groups of the window grid (NI * LH * W pixels)
  int ntiles, nslabs, ncols, nworkers, S;
};

__device__ __forceinline__ float xd_act(float a, int in_relu, int AM) {
  if (AM == ACT_RELU6) return fminf(fmaxf(a, 0.f), 6.f);
  if (AM == ACT_SWISH) return swish_f(a);
  if (AM == ACT_RELU) return fmaxf(a, 0.f);
  return in_relu ? fmaxf(a, 0.f) : a;
}
__device__ __forceinline__ float xd_act_bwd(float c, float a, int in_relu, int AM) {
  if (AM == ACT_RELU6) return (a > 0.f && a < 6.f) ? c : 0.f;
  if (AM == ACT_SWISH) return c * swish_grad(a);
  return (in_relu && !(a > 0.f)) ? 0.f : c;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float xd_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes, valid in lane 63, fixed order (see cw_wave_sum63 in dwconv_cw.hip)
__device__ __forceinline__ float xd_wave_sum63(float v) {
  v += xd_dpp<0xB1, 0xF>(v);
  v += xd_dpp<0x4E, 0xF>(v);
  v += xd_dpp<0x141, 0xF>(v);
  v += xd_dpp<0x140, 0xF>(v);
  v += xd_dpp<0x142, 0xA>(v);
  v += xd_dpp<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ unsigned xd_lds_addr(const void* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}
// operands of one tap row: K tap pairs as scalar loads, NR operand pairs from LDS (see cw_row_issue in dwconv_cw.hip)
template <int K, int NR>
__device__ __forceinline__ void xd_row_issue(f32x2 (&wr)[K], f32x2 (&v)[NR], const float* wp, unsigned tap_off, unsigned ld4, unsigned addr) {
#pragma unroll
  for (int kx = 0; kx < K; ++kx) asm volatile("s_load_dwordx2 %0, %1, %2" : "=&s"(wr[kx]) : "s"(wp), "s"(tap_off + (unsigned)kx * ld4));
#pragma unroll
  for (int i = 0; i < NR; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "i"(i * 8));
}
__device__ __forceinline__ void xd_row_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned xd_pack_bf16(float a, float b) {
  bf16x2 t;
  t[0] = (bf16_t)a; t[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float xd_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float xd_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// Workgroup -> (column of slabs, worker).  Blocks are placed on the XCDs round robin (block b on XCD b % 8): the columns of one
// worker (the same pixel tiles) are given to the SAME XCD, so its L2 fetches the tiles of x once for all columns.
__device__ __forceinline__ bool xd_block(const XdGeom& g, int& col, int& worker) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  col = idx % g.ncols;
  worker = (idx / g.ncols) * 8 + xcd;
  return worker < g.nworkers;
}

// The window grid of a tile in 16-pixel groups: group gr holds window pixels 16 gr .. 16 gr + 15 of (image, window row, column);
// wave wv of 8 owns the groups wv, wv + 8, ...  Per lane pixel j of its group gi, tile-independent and packed into one register:
// LDS byte address of the lane's first plane element (16 bits; the wave's dump slot for a lane beyond the window grid) |
// window row (5) << 16 | image (4) << 21 | column (6) << 25 | inside the window grid << 31.
template <int GPW>
struct XdGroups {
  unsigned d[GPW];
};
__device__ __forceinline__ unsigned xg_addr(unsigned d) { return d & 0xffffu; }
__device__ __forceinline__ int xg_wr(unsigned d) { return (d >> 16) & 31; }
__device__ __forceinline__ int xg_im(unsigned d) { return (d >> 21) & 15; }
__device__ __forceinline__ int xg_col(unsigned d) { return (d >> 25) & 63; }
__device__ __forceinline__ bool xg_in(unsigned d) { return (d >> 31) & 1; }
template <int GPW, int P>
__device__ __forceinline__ void xd_groups(XdGroups<GPW>& G, const XdGeom& g, int wv, int j, unsigned plane0_addr, unsigned dump_addr) {
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int qpx = (gi * 8 + wv) * 16 + j;
    const int per = g.LH * g.W;
    const int im = qpx / per, rem = qpx - im * per;
    const int wr = rem / g.W, c = rem - wr * g.W;
    const bool inwin = qpx < g.NI * per;
    const int woff = im * g.LH * g.LWp + wr * g.LWp + c + P;
    G.d[gi] = inwin ? ((plane0_addr + (unsigned)woff * 8u) | ((unsigned)wr << 16) | ((unsigned)im << 21) | ((unsigned)c << 25) | (1u << 31)) : dump_addr;
  }
}

// B fragments (16 pixels x 32 KC channels of x per group) of a tile's window grid, straight from global memory through a buffer
// resource over x: one add + one load per group.  A lane whose pixel lies before / behind the tensor is out of range and reads
// zeros; a pixel of a neighbouring image (halo rows across an image border) or the channels of the next pixel (inp < 32 KC) read
// finite values that are replaced by zeros in the window / meet zero weight columns.
template <int GPW, int KC, int P>
__device__ __forceinline__ void xd_load_b(bf16x8 (&bfr)[GPW][KC], const XdGroups<GPW>& G, const XdGeom& g, __amdgpu_buffer_rsrc_t rx, int ldx,
                                          int n0, int ho0, int q, int lane_off, int wv, int j) {
  const int tile_off = (int)((((long)n0 * g.H + ho0 - P) * g.W) * ldx * 2);   // bytes; negative above the first image
  const int pitch = ldx * 2;
  unsigned vo[GPW];
  if (g.NI == 1) {
    // row tiles: the window grid is one contiguous pixel range of x, group gi of this wave 128 pixels behind group gi - 1
    const int npx = g.LH * g.W;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int qpx = (gi * 8 + wv) * 16 + j;
      vo[gi] = ((gi * 8 + wv) * 16 + 16 <= npx || qpx < npx) ? (unsigned)(tile_off + lane_off + gi * 128 * pitch) : 0x80000000u;
    }
  } else {
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const unsigned d = G.d[gi];
      const int xoff = (xg_im(d) * g.H + xg_wr(d)) * g.W + xg_col(d);
      vo[gi] = xg_in(d) ? (unsigned)(tile_off + 16 * q + xoff * pitch) : 0x80000000u;
    }
  }
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      bfr[gi][kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, vo[gi] + (unsigned)(kc * 64), 0, 0));
}

// ---------------------------------------------------------------------------------------------------------------- forward
//   y = dwconv(act(scale * (x W1^T) + shift)),  stats: sum y, sum y^2 (of the stored values)
// One workgroup = 8 waves.  MFMA stage: the waves share the tile's window grid by 16-pixel groups and produce all 8 channel-pair
// planes of the slab; tap stage: wave wv is channel pair wv of the slab, its lanes are 7-pixel strips (k_dwf_cw).
//   KC   : 32-channel chunks of the block input (inp <= 32 KC)
//   GPW  : window groups per wave the instance is compiled for
//   KEEP : the B fragments (pixels of x) stay in registers across the slabs of a column; otherwise they are reloaded per slab
//          (L1 / L2 hits) because the tap loop needs the registers
//   PF   : two operand-row buffers in the tap loop (row ky + 1 in flight while row ky is multiplied)
constexpr int XD_SMAX = 3;   // slabs per column at most (statistics accumulators per slab live in registers)
template <int K, int AM, int KC, int GPW, bool KEEP, bool PF>
__global__ __launch_bounds__(512, 4) void k_xdwf(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ wexp, int ldwe,
                                                 const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_relu,
                                                 const float* __restrict__ w, int ldw, bf16_t* __restrict__ y, long yss,
                                                 float* __restrict__ stats, int stat_ld, int stat_rows, XdGeom g) {
  constexpr int P = (K - 1) / 2, SW = 7, IWN = SW + K - 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* s_in = reinterpret_cast<f32x2*>(smem);                        // [8 pairs][plane]: activated expand output of the window, fp32
  unsigned* s_y = reinterpret_cast<unsigned*>(s_in + 8 * g.plane);     // [8 pairs][TPIXp]: the tile's output (bf16 pairs)
  float* s_cf = reinterpret_cast<float*>(s_y + 8 * g.TPIXp);           // [XD_SMAX][2][16] scale / shift of the column's slabs
  f32x4* s_st = reinterpret_cast<f32x4*>(s_cf + XD_SMAX * 32);         // [XD_SMAX][8 waves][16 quads] statistics partials

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  int col, worker;
  if (!xd_block(g, col, worker)) return;
  const int slab0 = col * g.S;
  const int nsl = min(g.S, g.nslabs - slab0);   // slabs of this column (workgroup-uniform)

  for (int i = tid; i < 8 * g.plane; i += 512) s_in[i] = f32x2{0.f, 0.f};   // halo columns stay zero for the whole kernel
  if (tid < XD_SMAX * 32) {
    const int s = tid >> 5, v = (tid >> 4) & 1, c = (slab0 + s) * 16 + (tid & 15);
    s_cf[tid] = (s < nsl && c < g.C) ? (v == 0 ? in_scale[c] : in_shift[c]) : 0.f;
  }
  if (tid < XD_SMAX * 128) s_st[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

  XdGroups<GPW> G;
  xd_groups<GPW, P>(G, g, wv, j, xd_lds_addr(s_in + (2 * q) * g.plane), 0u);

  // tap stage work item of this lane: (image, row, strip)
  const int ipi = g.TH * g.ns;
  const int it_im = lane / ipi, it_rem = lane % ipi;
  const int it_r = it_rem / g.ns, it_j = it_rem % g.ns;
  const bool it_ok = lane < g.NI * ipi;
  const int pix0 = (it_im * g.TH + it_r) * g.W + SW * it_j;
  const unsigned in_addr0 = xd_lds_addr(s_in + wv * g.plane + it_im * g.LH * g.LWp + it_r * g.LWp + SW * it_j);

  // store slots: 16-byte piece p = tid + 512 i = (pixel pp of the tile, channel group cg = p & 1 of the slab);
  // packed: pp (10 bits) | row (5) << 10 | image (4) << 15 | exists << 19, and the element offset inside the slab
  unsigned sp_d[2];
  int sp_goff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + i * 512;
    const int pp = p >> 1, cg = p & 1;
    const int colx = pp % g.W, t2 = pp / g.W;
    const int rr = t2 % g.TH, im = t2 / g.TH;
    sp_d[i] = pp < g.TPIX ? ((unsigned)pp | ((unsigned)rr << 10) | ((unsigned)im << 15) | (1u << 19)) : 0u;
    sp_goff[i] = ((im * g.H + rr) * g.W + colx) * 16 + cg * 8;
  }

  const int t_beg = (int)((long)worker * g.ntiles / g.nworkers), t_end = (int)((long)(worker + 1) * g.ntiles / g.nworkers);
  unsigned ld4 = (unsigned)ldw * 4u;

  // ---- main loop over the (tile, slab) items of this workgroup.  Every vector-memory operation inside it is unconditional (loads
  // with clamped addresses, buffer stores whose invalid lanes are out of range): with a branch around any of them the compiler
  // cannot count the operations issued behind a prefetch and waits with vmcnt(0) -- i.e. for the output stores of the previous
  // item to be acknowledged -- before the fragments are used (25 % of the wave cycles + as much barrier skew, XD_TIMING build).
  // Order inside an item: MFMA stage (fragments fetched one item ago) -> prefetch of the next item's fragments -> stores of the
  // PREVIOUS item's output (s_y) -> barrier -> tap stage -> barrier.
  bf16x8 bfr[GPW][KC];
  bf16x8 afn[KC];   // A fragments (expand weights, row = channel of the slab) of the next MFMA stage
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, (int)((long)g.N * g.H * g.W * ldx * 2), 0x00020000);
  const int lane_off = (wv * 16 + j) * ldx * 2 + 16 * q;   // byte offset of this lane's pixel / channel quarter inside a row-tile window
  const int nitems = (t_end - t_beg) * nsl;
  int tile = t_beg, sl = 0;
  int nb = tile / g.tiles_y, ty = tile % g.tiles_y;
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) afn[kc] = *reinterpret_cast<const bf16x8*>(wexp + (long)(slab0 * 16 + j) * ldwe + kc * 32 + 8 * q);
  if (KEEP) xd_load_b<GPW, KC, P>(bfr, G, g, rx, ldx, nb * g.NI, ty * g.TH, q, lane_off, wv, j);
  // the first fragments are waited for here, so that the loop is entered with nothing pending
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) asm volatile("" : "+v"(afn[kc]));
  if (KEEP) {
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) asm volatile("" : "+v"(bfr[gi][kc]));
  }
  __syncthreads();   // planes zeroed, coefficients staged

  unsigned p_off = 0x80000000u;   // byte offset of the previous item's tile inside its slab (out of range: nothing to store yet)
  int p_slab = slab0, p_n0 = 0, p_ho0 = 0;
  unsigned vmask = 0;
#if XD_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
#pragma unroll 1
  for (int it = 0; it < nitems; ++it) {
    const int n0 = nb * g.NI, ho0 = ty * g.TH;
    const int c_base = (slab0 + sl) * 16;
    // opaque per item: the window / store addresses are formed where they are used (hoisted out of the loop they are spilled)
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) asm volatile("" : "+v"(G.d[gi]));
    asm volatile("" : "+v"(sp_d[0]), "+v"(sp_d[1]), "+v"(sp_goff[0]), "+v"(sp_goff[1]));
    if (sl == 0) {
      // validity of this lane's window pixels in this tile (image borders, ragged batch): one bit per group, for all slabs
      vmask = 0;
#pragma unroll
      for (int gi = 0; gi < GPW; ++gi) {
        const unsigned d = G.d[gi];
        const bool valid = xg_in(d) && n0 + xg_im(d) < g.N && (unsigned)(ho0 - P + xg_wr(d)) < (unsigned)g.H;
        vmask |= valid ? 1u << gi : 0u;
      }
    }
    // ---- MFMA stage: planes of slab sl for the whole window
    bf16x8 afr[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) afr[kc] = afn[kc];
    if (!KEEP) xd_load_b<GPW, KC, P>(bfr, G, g, rx, ldx, n0, ho0, q, lane_off, wv, j);
#if XD_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XDMARK(6)
#endif
    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(s_cf + sl * 32 + 4 * q);
    const f32x4 sh4 = *reinterpret_cast<const f32x4*>(s_cf + sl * 32 + 16 + 4 * q);
    f32x2 sc01 = f32x2{sc4[0], sc4[1]}, sc23 = f32x2{sc4[2], sc4[3]}, sh01 = f32x2{sh4[0], sh4[1]}, sh23 = f32x2{sh4[2], sh4[3]};
    // the coefficients are waited for ONCE, here: read first inside the per-group blocks, every block gets its own s_waitcnt lgkmcnt(0),
    // which then also waits for the window writes of the group before it (the stage serialises on the LDS write latency)
    asm volatile("" : "+v"(sc01), "+v"(sc23), "+v"(sh01), "+v"(sh23));
    // all MFMAs of the stage first (independent accumulators: their latency overlaps), then the epilogues
    f32x4 accs[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      accs[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((gi * 8 + wv) < g.NG) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) accs[gi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], bfr[gi][kc], accs[gi], 0, 0, 0);
      }
    }
    const unsigned plane_b = (unsigned)g.plane * 8u;
    const int npx_win = g.NI * g.LH * g.W;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if ((gi * 8 + wv) < g.NG) {
        const f32x4 acc = accs[gi];
        f32x2 a01 = f32x2{acc[0], acc[1]} * sc01 + sh01, a23 = f32x2{acc[2], acc[3]} * sc23 + sh23;
        a01[0] = xd_act(a01[0], in_relu, AM); a01[1] = xd_act(a01[1], in_relu, AM);
        a23[0] = xd_act(a23[0], in_relu, AM); a23[1] = xd_act(a23[1], in_relu, AM);
        const bool valid = (vmask >> gi) & 1u;
        const f32x2 z = f32x2{0.f, 0.f};
        a01 = valid ? a01 : z; a23 = valid ? a23 : z;
        // only the group that straddles the end of the window grid has lanes without a window pixel (wave-uniform test)
        const unsigned a0 = xg_addr(G.d[gi]);
        if (!(XD_KO & 4) && ((gi * 8 + wv) * 16 + 16 <= npx_win || xg_in(G.d[gi]))) {
          typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
          *reinterpret_cast<lds_f32x2*>((size_t)a0) = a01;
          *reinterpret_cast<lds_f32x2*>((size_t)(a0 + plane_b)) = a23;
        }
      }
    }
    XDMARK(7)
    // ---- next item; its fragments are fetched behind the tap stage (the last item fetches its own again)
    int nsl_ = sl + 1, nnb = nb, nty = ty;
    if (nsl_ == nsl) {
      nsl_ = 0;
      if (it + 1 < nitems) { ++nty; if (nty == g.tiles_y) { nty = 0; ++nnb; } }
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      afn[kc] = *reinterpret_cast<const bf16x8*>(wexp + (long)((slab0 + nsl_) * 16 + j) * ldwe + kc * 32 + 8 * q);
    if (KEEP && !(XD_KO & 8)) xd_load_b<GPW, KC, P>(bfr, G, g, rx, ldx, nnb * g.NI, nty * g.TH, q, lane_off, wv, j);
    XDMARK(0)
    // ---- the previous item's output leaves s_y (complete since barrier (2)); a piece that does not exist is out of range
    {
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + (long)p_slab * yss, 0, (int)0x80000000u, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned d = sp_d[i];
        const int pp = d & 0x3ff, rr = (d >> 10) & 31, im = (d >> 15) & 15;
        const bool ok = (d >> 19) && p_n0 + im < g.N && p_ho0 + rr < g.H;
        const int cg = (tid + i * 512) & 1;
        xu32x4 v;
        const unsigned* sy_ = s_y + (cg * 4) * g.TPIXp + pp;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) v[qq] = sy_[qq * g.TPIXp];
        if (!(XD_KO & 1)) __builtin_amdgcn_raw_buffer_store_b128(v, ry, ok ? p_off + (unsigned)sp_goff[i] * 2u : 0x80000000u, 0, 0);
      }
    }
    XDMARK(4)
    if (!(XD_KO & 16)) __syncthreads();   // (1) planes complete; the previous item's output has left s_y
    XDMARK(1)

    // ---- tap stage: wave = channel pair
    const int ch = c_base + 2 * wv;
    const float* wp = w + ch;
    asm volatile("" : "+s"(ld4));
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (!(XD_KO & 2) && it_ok && n0 + it_im < g.N && ho0 + it_r < g.H && ch < g.C) {
      f32x2 acc[SW];
#pragma unroll
      for (int t = 0; t < SW; ++t) acc[t] = f32x2{0.f, 0.f};
      f32x2 inb[PF ? 2 : 1][IWN], wb[PF ? 2 : 1][K];
      if (PF) xd_row_issue<K, IWN>(wb[0], inb[0], wp, 0u, ld4, in_addr0);
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int cur = PF ? (ky & 1) : 0;
        if (!PF) xd_row_issue<K, IWN>(wb[0], inb[0], wp, (unsigned)(ky * K) * ld4, ld4, in_addr0 + (unsigned)(ky * g.LWp) * 8u);
        xd_row_wait();
        if (PF && ky + 1 < K) {
          xd_row_issue<K, IWN>(wb[cur ^ 1], inb[cur ^ 1], wp, (unsigned)((ky + 1) * K) * ld4, ld4,
                               in_addr0 + (unsigned)((ky + 1) * g.LWp) * 8u);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int t = 0; t < SW; ++t) acc[t] += inb[cur][t + kx] * wb[cur][kx];
        }
#pragma unroll
        for (int t = 0; t < SW; ++t) asm volatile("" : "+v"(acc[t]));
        __builtin_amdgcn_sched_barrier(0);
      }
      unsigned* yp = s_y + wv * g.TPIXp + pix0;
      const bool ch1_ok = ch + 1 < g.C;
#pragma unroll
      for (int t = 0; t < SW; ++t) {
        const unsigned o = xd_pack_bf16(acc[t][0], ch1_ok ? acc[t][1] : 0.f);
        const float v0 = xd_lo(o), v1 = xd_hi(o);
        t0 += v0; t1 += v1; t2 += v0 * v0; t3 += v1 * v1;
        yp[t] = o;
      }
    }
    // statistics: quad sums (2 DPP steps), then 16 lanes of the wave add into their LDS slot of (slab, wave) -- fixed order
    {
      t0 += xd_dpp<0xB1, 0xF>(t0); t1 += xd_dpp<0xB1, 0xF>(t1); t2 += xd_dpp<0xB1, 0xF>(t2); t3 += xd_dpp<0xB1, 0xF>(t3);
      t0 += xd_dpp<0x4E, 0xF>(t0); t1 += xd_dpp<0x4E, 0xF>(t1); t2 += xd_dpp<0x4E, 0xF>(t2); t3 += xd_dpp<0x4E, 0xF>(t3);
      if ((lane & 3) == 0) {
        f32x4* a = s_st + (sl * 8 + wv) * 16 + (lane >> 2);
        f32x4 v = *a;
        v[0] += t0; v[1] += t1; v[2] += t2; v[3] += t3;
        *a = v;
      }
    }
    XDMARK(2)
    if (!(XD_KO & 32)) __syncthreads();   // (2) output tile complete in s_y, window consumed
    XDMARK(3)
    p_off = (unsigned)((((long)n0 * g.H + ho0) * g.W * 16) * 2);
    p_slab = slab0 + sl; p_n0 = n0; p_ho0 = ho0;
    sl = nsl_; nb = nnb; ty = nty;
  }
  // the last item's output
  {
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + (long)p_slab * yss, 0, (int)0x80000000u, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned d = sp_d[i];
      const int pp = d & 0x3ff, rr = (d >> 10) & 31, im = (d >> 15) & 15;
      const bool ok = (d >> 19) && p_n0 + im < g.N && p_ho0 + rr < g.H && p_off != 0x80000000u;
      const int cg = (tid + i * 512) & 1;
      xu32x4 v;
      const unsigned* sy_ = s_y + (cg * 4) * g.TPIXp + pp;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) v[qq] = sy_[qq * g.TPIXp];
      __builtin_amdgcn_raw_buffer_store_b128(v, ry, ok ? p_off + (unsigned)sp_goff[i] * 2u : 0x80000000u, 0, 0);
    }
  }
#if XD_TIMING
  XDMARK(5)
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_xd_timing[i], tacc[i]);
  }
#endif

  if (stats) {
    // every wave adds its 16 quad partials per slab in order (only this wave touched them: no barrier needed)
    for (int s = 0; s < nsl; ++s) {
      if (lane < 2) {
        const int c = (slab0 + s) * 16 + 2 * wv + lane;
        if (c < g.C) {
          float sum = 0.f, sq = 0.f;
          for (int i = 0; i < 16; ++i) {
            const f32x4 v = s_st[(s * 8 + wv) * 16 + i];
            sum += lane ? v[1] : v[0];
            sq += lane ? v[3] : v[2];
          }
          float* r = stats + (long)worker * 2 * stat_ld;
          r[c] = sum;
          r[stat_ld + c] = sq;
          stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, c);
          stat_zero_tail(stats, 2L * stat_ld, worker + g.nworkers, g.nworkers, stat_rows, (long)stat_ld + c);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- Gram-matrix statistics
// Statistics of the expand BatchNorm without the expanded tensor: with e_c = w_c . x,
//   sum_m e_c = w_c . sx,   sum_m e_c^2 = w_c^T G w_c,     sx = sum_m x_m,  G = X^T X  (inp x inp)
// w_c = the bf16 weights the MFMA stage multiplies with; the quadratic form is accumulated in double.
// Output: ONE partial row [2][stat_ld] in the format of atomnas_bn_finalize_fwd (stat_rows = 1).
__global__ __launch_bounds__(64) void k_gram_stats(const float* __restrict__ gram, int ldg, const float* __restrict__ sx,
                                                   const bf16_t* __restrict__ wexp, int ldwe, int inp, int C, float* __restrict__ stats,
                                                   int stat_ld) {
  extern __shared__ float s_g[];        // [inp][inp] Gram matrix, [inp] sx, [64][inp + 1] weight rows
  float* s_sx = s_g + inp * inp;
  float* s_w = s_sx + inp;
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + tid;
  for (int i = tid; i < inp * inp; i += 64) s_g[i] = gram[(i / inp) * ldg + i % inp];
  for (int i = tid; i < inp; i += 64) s_sx[i] = sx[i];
  for (int i = tid; i < 64 * inp; i += 64) {
    const int r = i / inp, k = i - r * inp;
    const int cc = blockIdx.x * 64 + r;
    s_w[r * (inp + 1) + k] = cc < C ? (float)wexp[(long)cc * ldwe + k] : 0.f;
  }
  __syncthreads();
  if (c >= C) return;
  const float* wr = s_w + tid * (inp + 1);
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < inp; ++i) {
    double t = 0.0;
    for (int jj = 0; jj < inp; ++jj) t += (double)s_g[i * inp + jj] * (double)wr[jj];
    s2 += (double)wr[i] * t;
    s1 += (double)wr[i] * (double)s_sx[i];
  }
  stats[c] = (float)s1;
  stats[stat_ld + c] = (float)(s2 > 0.0 ? s2 : 0.0);
}

// ---------------------------------------------------------------------------------------------------------------- host side
static bool xd_geometry(XdGeom& g, int N, int H, int W, int C, int inp, int K) {
  if (W % 7 != 0 || W < 7 || C % 16 != 0) return false;
  g.N = N; g.H = H; g.W = W; g.C = C; g.inp = inp;
  g.ns = W / 7;
  if (g.ns > 16) return false;
  if (H * g.ns <= 64) {   // whole images
    g.TH = H; g.tiles_y = 1; g.NI = 64 / (H * g.ns);
    if (g.NI > N) g.NI = N;
  } else {
    const int cap = 64 / g.ns;
    const int nty = (H + cap - 1) / cap;
    g.TH = (H + nty - 1) / nty;
    g.tiles_y = (H + g.TH - 1) / g.TH;
    g.NI = 1;
  }
  g.LH = g.TH + K - 1;
  if (g.LH > 31 || g.NI > 15 || W > 63 || g.NI * g.TH * W > 1023) return false;
  const int lw = W + K - 1;
  const bool pow2 = (g.ns & (g.ns - 1)) == 0;
  int lwp = lw;
  if (pow2) { while (lwp % (2 * g.ns) != g.ns) ++lwp; } else if (lwp % 2 == 0) ++lwp;
  g.LWp = lwp;
  // the 32 lanes of an LDS write group are 16 pixels x 2 quarters q of the MFMA tile, q's planes 2 * plane elements apart: with
  // plane = 8 (mod 16) the two quarters land on the two halves of the 64 banks
  int plane = g.NI * g.LH * g.LWp;
  while (plane % 16 != 8) ++plane;
  if (plane * 8 * 8 > 65535) return false;
  g.plane = plane;
  g.TPIX = g.NI * g.TH * W;
  int tp = g.TPIX;
  while (tp % 8 != 4) ++tp;
  g.TPIXp = tp;
  g.NG = (g.NI * g.LH * W + 15) / 16;
  g.ntiles = ((N + g.NI - 1) / g.NI) * g.tiles_y;
  g.nslabs = C / 16;
  return true;
}
static size_t xd_lds_fwd(const XdGeom& g) {
  return (size_t)8 * g.plane * sizeof(f32x2) + (size_t)8 * g.TPIXp * sizeof(unsigned) + 3 * 32 * sizeof(float) + 3 * 128 * sizeof(f32x4);
}
static void xd_workers(XdGeom& g, int per_cu, int max_rows) {
  static const int s_env = getenv("ATOMNAS_XDW_S") ? atoi(getenv("ATOMNAS_XDW_S")) : 3;   // experiment switch: slabs per column (1..3)
  g.S = s_env < 1 ? 1 : (s_env > 3 ? 3 : s_env);
  g.ncols = (g.nslabs + g.S - 1) / g.S;
  if (per_cu < 1) per_cu = 1;
  long want = ((long)num_cus() * per_cu) / g.ncols;
  static const long max_env = getenv("ATOMNAS_DW_MAX_WORKERS") ? atol(getenv("ATOMNAS_DW_MAX_WORKERS")) : 0;   // tests: long tile walks
  if (max_env > 0 && want > max_env) want = max_env;
  if (max_rows > 0 && want > max_rows) want = max_rows;   // every worker owns one partial row
  if (want > g.ntiles) want = g.ntiles;
  if (want >= 8) want -= want % 8;                        // whole rounds of XCDs
  if (want < 1) want = 1;
  g.nworkers = (int)want;
  static const bool dbg = getenv("ATOMNAS_XDW_DEBUG") != nullptr;   // experiment switch: launch geometry to stderr
  if (dbg) fprintf(stderr, "xdw_fwd: H %d C %d per_cu %d ncols %d S %d workers %d ntiles %d NG %d plane %d\n", g.H, g.C, per_cu, g.ncols, g.S, g.nworkers, g.ntiles, g.NG, g.plane);
}
static unsigned xd_grid(const XdGeom& g) { return (unsigned)(((g.nworkers + 7) / 8) * 8 * g.ncols); }

static int xd_gpw(const XdGeom& g) { return (g.NG + 7) / 8; }

// instance rules: fragments kept across the slabs wherever the registers allow it; k = 7 gives up the second operand-row buffer for them
#ifndef XD_KEEP
#define XD_KEEP(K, KC) ((K) == 3 || (KC) == 1)
#endif
#ifndef XD_PF
#define XD_PF(K, KC) ((KC) == 1 ? (K) < 7 : (K) > 3)
#endif
template <int K, int KC, int GPW>
static int xd_launch_fwd(const void* x, int ldx, const void* wexp, int ldwe, const float* sc, const float* sh, int relu, const float* w, int ldw,
                         void* y, long yss, float* stats, int stat_ld, int stat_rows, XdGeom g, hipStream_t st) {
  const size_t lds = xd_lds_fwd(g);
#define XD_FWD(AMV)                                                                                                          \
  {                                                                                                                          \
    auto kern = k_xdwf<K, AMV, KC, GPW, XD_KEEP(K, KC), XD_PF(K, KC)>;                                                                            \
    xd_workers(g, resident_per_cu(kern, 512, lds), stats ? stat_rows : 0);                                                   \
    hipLaunchKernelGGL(kern, dim3(xd_grid(g)), dim3(512), lds, st, (const bf16_t*)x, ldx, (const bf16_t*)wexp, ldwe, sc, sh, relu, w, ldw, \
                       (bf16_t*)y, yss, stats, stat_ld, stat_rows, g);                                                       \
  }
  if (relu == ACT_RELU6) XD_FWD(ACT_RELU6) else if (relu == ACT_SWISH) XD_FWD(ACT_SWISH) else XD_FWD(ACT_RELU)
#undef XD_FWD
  return check_launch("xdw_fwd");
}

static int xd_mode() {
  static const int m = getenv("ATOMNAS_XDW") ? atoi(getenv("ATOMNAS_XDW")) : 1;   // experiment switch: 0 = the fused kernels are not offered
  return m;
}

}  // namespace atomnas

using namespace atomnas;

// 1 when atomnas_xdw_fwd / atomnas_xdw_bwd have an instance for the shape: bf16, stride 1, inp <= 64 (a multiple of 8), C a multiple
// of 16, image width a multiple of 7, window groups per wave <= 7 and the LDS planes within 160 KB.
extern "C" int atomnas_xdw_supported(int N, int H, int W, int inp, int C, int k, int stride, int dtype) {
  if (!xd_mode() || dtype != DT_BF16 || stride != 1 || !(k == 3 || k == 5 || k == 7) || inp < 8 || inp > 64 || inp % 8) return 0;
  XdGeom g;
  if (!xd_geometry(g, N, H, W, C, inp, k)) return 0;
  if (xd_gpw(g) > (inp <= 32 ? 7 : 5)) return 0;
  if (!xdw_cw_bwd_supported(N, H, W, C, k)) return 0;
  return xd_lds_fwd(g) <= max_lds_bytes() ? 1 : 0;
}

// Forward of expand 1x1 + BatchNorm + activation + depthwise k x k of one branch segment (models/mobilenet_base.py:316-336):
//   y[M][C] (slab-major) = dwconv_k(act(in_scale * (x W^T) + in_shift)),  stats rows: sum y, sum y^2 of the stored values.
// x [M = N H W][ldx] plain bf16 (the block input); wexp: the segment's rows of the packed expand weight ([C][ldwe >= inp rounded up
// to 32], zero padded); in_scale / in_shift: the expand BatchNorm's coefficients of the segment; w: depthwise taps [k*k][ldw] fp32.
extern "C" int atomnas_xdw_fwd(const void* x, int ldx, int inp, const void* wexp, int ldwe, const float* in_scale, const float* in_shift,
                               int act, const float* w, int ldw, void* y, long y_ss, float* stats, int stat_ld, int stat_rows, int N, int H,
                               int W, int C, int k, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && wexp && in_scale && in_shift && w && y, "xdw_fwd: null pointer");
  ATOMNAS_REQUIRE(act == ACT_RELU || act == ACT_RELU6 || act == ACT_SWISH, "xdw_fwd: activation mode %d (the expand ConvBNReLU has one: 1..3)", act);
  ATOMNAS_REQUIRE(atomnas_xdw_supported(N, H, W, inp, C, k, 1, dtype), "xdw_fwd: unsupported shape N=%d H=%d W=%d inp=%d C=%d k=%d", N, H, W, inp, C, k);
  ATOMNAS_REQUIRE(ldx >= inp && ldx % 8 == 0 && ldwe >= (inp + 31) / 32 * 32 && ldwe % 8 == 0 && ldw >= C && y_ss >= (long)N * H * W * 16,
                  "xdw_fwd: bad pitch (ldx=%d ldwe=%d ldw=%d)", ldx, ldwe, ldw);
  ATOMNAS_REQUIRE(!stats || (stat_ld >= C && stat_rows > 0), "xdw_fwd: statistics pitch %d < C=%d or stat_rows=%d", stat_ld, C, stat_rows);
  ATOMNAS_REQUIRE((long)N * H * W * ldx * 2 < (1L << 31) && (long)N * H * W * 32 < (1L << 31), "xdw_fwd: tensor beyond 2 GB per buffer resource");
  XdGeom g;
  xd_geometry(g, N, H, W, C, inp, k);
  hipStream_t st = (hipStream_t)stream;
  const int kc = (inp + 31) / 32;
#define XD_CASE(KV, KCV) \
  if (k == KV && kc == KCV) return xd_launch_fwd<KV, KCV, (KCV == 1 ? 7 : 5)>(x, ldx, wexp, ldwe, in_scale, in_shift, act, w, ldw, y, y_ss, stats, stat_ld, stat_rows, g, st);
  XD_CASE(3, 1) XD_CASE(5, 1) XD_CASE(7, 1) XD_CASE(3, 2) XD_CASE(5, 2) XD_CASE(7, 2)
#undef XD_CASE
  set_error("xdw_fwd: no instance");
  return 1;
}

// Statistics row of the expand BatchNorm from the Gram matrix of the block input (see k_gram_stats).  gram [inp][ldg], sx [inp]:
// atomnas_gram's outputs; wexp: packed expand weight [C][ldwe] bf16.  Writes stats[0 .. C) = sum e_c and stats[stat_ld .. + C) = sum e_c^2.
extern "C" int atomnas_gram_stats(const float* gram, int ldg, const float* sx, const void* wexp, int ldwe, int inp, int C, float* stats,
                                  int stat_ld, void* stream) {
  ATOMNAS_REQUIRE(gram && sx && wexp && stats && inp > 0 && inp <= 64 && C > 0 && ldg >= inp && ldwe >= inp && stat_ld >= C,
                  "gram_stats: bad arguments");
  const size_t lds = ((size_t)inp * inp + inp + 64 * (inp + 1)) * sizeof(float);
  hipLaunchKernelGGL(k_gram_stats, dim3((C + 63) / 64), dim3(64), lds, (hipStream_t)stream, gram, ldg, sx, (const bf16_t*)wexp, ldwe, inp, C,
                     stats, stat_ld);
  return check_launch("gram_stats");
}

// Backward of the depthwise convolution of one branch segment with its input operand recomputed from the block input
// (atomnas_dwconv_bwd's arithmetic with e = x W^T in place of the stream `x`; models/mobilenet_base.py:316-336 backward):
//   dYraw = c1*g + c2*yraw + c3;  h = dwconv^T(dYraw) * act'(in_scale*e + in_shift)  -> h (slab-major, storage type);
//   dw[C][k*k] += corr(act(in_scale*e + in_shift), dYraw);  stats rows: sum h, sum h*e.
extern "C" int atomnas_xdw_bwd(const void* g, long g_ss, const void* yraw, long yraw_ss, const float* c1, const float* c2, const float* c3,
                               const void* x, int ldx, int inp, const void* wexp, int ldwe, const float* in_scale, const float* in_shift,
                               int act, const float* w, int ldw, void* h, long h_ss, float* dw, float* stats, int stat_ld, int part_rows,
                               float* dw_ws, int N, int H, int W, int C, int k, int dtype, void* stream) {
  ATOMNAS_REQUIRE(g && x && wexp && in_scale && in_shift && w && h, "xdw_bwd: null pointer");
  ATOMNAS_REQUIRE(atomnas_xdw_supported(N, H, W, inp, C, k, 1, dtype), "xdw_bwd: unsupported shape N=%d H=%d W=%d inp=%d C=%d k=%d", N, H, W, inp, C, k);
  const long M = (long)N * H * W;
  ATOMNAS_REQUIRE(ldx >= inp && ldx % 8 == 0 && ldwe >= (inp + 31) / 32 * 32 && ldwe % 8 == 0 && ldw >= C && g_ss >= M * 16 && h_ss >= M * 16,
                  "xdw_bwd: bad pitch (ldx=%d ldwe=%d ldw=%d)", ldx, ldwe, ldw);
  ATOMNAS_REQUIRE(!yraw || (c1 && c2 && c3 && yraw_ss >= M * 16), "xdw_bwd: yraw needs c1,c2,c3 and a valid slab stride");
  ATOMNAS_REQUIRE(!stats || stat_ld >= C, "xdw_bwd: statistics pitch %d < C=%d", stat_ld, C);
  ATOMNAS_REQUIRE(!(stats || dw) || part_rows > 0, "xdw_bwd: part_rows must be positive");
  ATOMNAS_REQUIRE(!dw || dw_ws, "xdw_bwd: the weight gradient needs the partial workspace dw_ws [part_rows][C][k*k]");
  const int rc = xdw_cw_bwd(g, g_ss, yraw, yraw_ss, c1, c2, c3, x, ldx, inp, wexp, ldwe, in_scale, in_shift, act, w, ldw, h, h_ss, dw, stats,
                            stat_ld, part_rows, dw_ws, N, H, W, C, k, (hipStream_t)stream);
  if (rc < 0) { set_error("xdw_bwd: no instance"); return 1; }
  return rc;
}

#if XD_TIMING
extern "C" int atomnas_debug_xd_timing(unsigned long long* out8, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(atomnas::g_xd_timing), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(atomnas::g_xd_timing), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif

// Shared pieces of the "one channel pair per wave" depthwise kernels (gfx950): tile geometry, storage-type plumbing, staging-slot
// decode, workgroup -> (slab, worker, half) mapping, launch sizing.  Included by dwconv_cw.hip (packed-FMA tap rows), dwconv_mm.hip
// (tap rows on the matrix cores) and the round-4 experiment csrc/experimental/xdw_cw_bwd.hip.  See dwconv_cw.hip for the design.
#pragma once
#include "common.h"
#include <cstdlib>

#ifndef CW_TIMING
#define CW_TIMING 0   // experiment builds (tools/variant.sh): s_memtime accounting of the phases of a tile in the backward kernel
#endif

namespace atomnas {

#if CW_TIMING
#define CWMARK(i)                                                    \
  {                                                                  \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long tn_ = __builtin_readcyclecounter();     \
    tacc[i] += tn_ - tlast;                                          \
    tlast = tn_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                               \
  }
#else
#define CWMARK(i)
#endif

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct CwGeom {
  int N, H, W, C;
  int TH, NI, tiles_y, ns;   // tile: NI images x TH rows x W columns; ns = W / 7 strips per row
  int LH, LWp, plane;        // operand window: rows per image, row pitch, elements (f32x2) per channel-pair plane
  int RH;                    // rows of the window ring per image (= LH)
  int TPIX, TPIXp;           // pixels per tile; pitch of the pixel planes
  int nworkers, nslabs, ntiles;
  int ring;                  // 1: several tiles per image, the window rows are a ring
  // stride-2 kernels (k_dwb_cw2 / k_dwf_cw2): the lanes live on the dY / output grid (Ho x Wo), the window holds that grid
  int Ho, Wo, THd;           // output rows / columns, output rows per tile (TH = 2 THd input rows)
  int TPIXD;                 // output pixels per tile
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
  static __device__ __forceinline__ void touch(const piece_t& p) { asm volatile("" ::"v"(p.v)); }   // "the register is read here"
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
  static __device__ __forceinline__ void touch(const piece_t& p) { asm volatile("" ::"v"(p.a), "v"(p.b)); }
};

__device__ __forceinline__ float cw_act(float a, int in_relu, int AM) {
  if (AM == ACT_RELU) return fmaxf(a, 0.f);   // (instances compiled for the ReLU mode: no run-time select per value)
  if (AM == ACT_RELU6) return fminf(fmaxf(a, 0.f), 6.f);
  if (AM == ACT_SWISH) return swish_f(a);
  return in_relu ? fmaxf(a, 0.f) : a;
}
__device__ __forceinline__ float cw_act_bwd(float c, float a, int in_relu, int AM) {
  if (AM == ACT_RELU) return a > 0.f ? c : 0.f;
  if (AM == ACT_RELU6) return (a > 0.f && a < 6.f) ? c : 0.f;
  if (AM == ACT_SWISH) return c * swish_grad(a);
  return (in_relu && !(a > 0.f)) ? 0.f : c;
}

// Activation followed by the clamp to the fp16 range that the matrix-core forward kernels apply to their operand (dwconv_mm.hip,
// dwconv_mm2.hip): ONE v_med3_f32 for none / ReLU / ReLU6 (round 6: was v_max + v_med3 per element of the commit phase, the phase those
// kernels spend their issue slots in; same values for every non-NaN input).
// `ok` (the piece lies inside the image; one flag for its 8 channels) rides in the bounds: an invalid piece is clamped to [0, 0], which
// replaces a select per element by two per piece.  (The values of an invalid piece come from a clamped address: finite.)
struct CwClamp { float lo, hi; };
__device__ __forceinline__ CwClamp cw_clamp16_bounds(bool ok, int in_relu, int AM) {
  const float lo = (AM == ACT_RELU || AM == ACT_RELU6 || (AM == 0 && in_relu)) ? 0.f : -65504.f;
  const float hi = AM == ACT_RELU6 ? 6.f : 65504.f;
  return CwClamp{ok ? lo : 0.f, ok ? hi : 0.f};
}
__device__ __forceinline__ float cw_act_clamp16(float a, CwClamp b, int AM) {
  if (AM == ACT_SWISH) a = swish_f(a);
  return __builtin_amdgcn_fmed3f(a, b.lo, b.hi);
}

// Sum over the 64 lanes of a wave with DPP adds only (no LDS traffic): quad butterflies, half-row and row mirrors leave every
// lane with the sum of its 16-lane row; row_bcast15 / row_bcast31 then carry the row sums upwards.  The total is valid in
// lanes 48..63 (lane 63 is read); the order of the additions is fixed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float cw_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float cw_wave_sum63(float v) {
  v += cw_dpp<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += cw_dpp<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += cw_dpp<0x141, 0xF>(v);   // row_half_mirror
  v += cw_dpp<0x140, 0xF>(v);   // row_mirror
  v += cw_dpp<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v += cw_dpp<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ unsigned cw_lds_addr(const void* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}
// The operands of one tap row, all asynchronous: the K tap pairs of the wave's channel pair as scalar loads (s_load_dwordx2 from
// the tap-major table: wave-uniform, so they cost no vector instruction and no vector register -- v_readlane broadcasts were
// measured at 9 cycles each, 14 per row next to 98 FMAs of ~4, tools/probe/vpk.hip) and the lane's NR consecutive operand pairs
// from LDS.  Inline asm on purpose: hipcc's load/store optimizer merges neighbouring ds_read_b64 into ds_read2_b64, which moves
// the same bytes in twice the LDS cycles (MI355X_MICROARCH.md, LDS table), and it cannot keep SMEM results in flight across
// its own waits.  The loads are invisible to the compiler's wait counters: cw_row_wait() makes every result valid before its first
// use (s_waitcnt + scheduling fence; cdna_hip_programming.md 5.7 form iii).
template <int K, int NR>
__device__ __forceinline__ void cw_row_issue(f32x2 (&wr)[K], f32x2 (&v)[NR], const float* wp, unsigned tap_off, unsigned ld4, unsigned addr) {
#pragma unroll
  for (int kx = 0; kx < K; ++kx) asm volatile("s_load_dwordx2 %0, %1, %2" : "=&s"(wr[kx]) : "s"(wp), "s"(tap_off + (unsigned)kx * ld4));
#pragma unroll
  for (int i = 0; i < NR; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "i"(i * 8));
}
__device__ __forceinline__ void cw_row_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Tile-independent decode of the two staging slots of a thread: slot i of thread tid is 16-byte piece (tid + i * NT) of the tile;
// pieces run over (image, row, column, channel group of the workgroup) with the channel group fastest.
struct CwSlots {
  int pp[2];     // pixel index inside the tile (im, row, col) -> also the index into the pixel planes; -1: no such piece
  int rr[2];     // row inside the tile
  int dyo[2];    // window element offset without the row term: im * LH * LWp + col + P
  int goff[2];   // element offset inside the slab relative to the tile's first pixel: ((im * H + rr) * W + col) * 16 + cg * 8
  int im[2];
};
template <int P, int NT, int CGS>
__device__ __forceinline__ void cw_decode(CwSlots& s, const CwGeom& g, int tid, int cg) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = (tid + i * NT) / CGS;
    const bool ok = pp < g.TPIX;
    const int col = pp % g.W, t2 = pp / g.W;
    const int rr = t2 % g.TH, im = t2 / g.TH;
    s.pp[i] = ok ? pp : -1;
    s.rr[i] = rr;
    s.im[i] = im;
    s.dyo[i] = im * g.RH * g.LWp + col + P;
    s.goff[i] = ((im * g.H + rr) * g.W + col) * 16 + cg * 8;
  }
}

// Workgroup -> (slab, worker, half).  NW = 8: the 8 waves are the 8 channel pairs of a slab.  NW = 4: a workgroup owns 8 of the 16
// channels (the 16-byte half of every 32-byte pixel); blocks b and b + 8 -- the same XCD, i.e. the same L2, under the observed
// round-robin placement -- are the two halves of one (slab, worker), so the shared 128-byte lines are fetched from HBM once.
template <int NW>
__device__ __forceinline__ bool cw_block(const CwGeom& g, int& slab, int& worker, int& half) {
  if (NW == 8) {
    slab = blockIdx.x % g.nslabs; worker = blockIdx.x / g.nslabs; half = 0;
    return true;
  }
  const int q = blockIdx.x >> 3;
  half = q & 1;
  const int u = (q >> 1) * 8 + (blockIdx.x & 7);
  slab = u % g.nslabs; worker = u / g.nslabs;
  return worker < g.nworkers;
}

constexpr __host__ __device__ int cw_fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

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
  g.RH = g.LH;
  int plane = g.NI * g.RH * g.LWp;
  if (plane < 512) plane = 512;      // the weight-gradient flush transposes 64 x 15 + 56 floats through a wave's own plane
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

// stride 2: the lane grid is the output grid; K decides the halo rows / columns of the window
static bool cw2_geometry(CwGeom& g, int N, int H, int W, int C, int K) {
  if (H % 2 || W % 14 != 0 || W < 14) return false;
  const int P = (K - 1) / 2;
  const int RELMIN = cw_fdiv(-P, 2), RELMAX = cw_fdiv(13 + P, 2), CL = -RELMIN, CR = RELMAX - 6;
  const int HL = P / 2 + cw_fdiv(P - 1, 2) + 1;
  g.N = N; g.H = H; g.W = W; g.C = C;
  g.Ho = H / 2; g.Wo = W / 2;
  g.ns = g.Wo / 7;
  if (g.ns > 16) return false;
  if (g.Ho * g.ns <= 64) {   // whole images
    g.THd = g.Ho; g.tiles_y = 1; g.NI = 64 / (g.Ho * g.ns); g.ring = 0;
    if (g.NI > N) g.NI = N;
  } else {
    const int cap = 64 / g.ns;
    const int nty = (g.Ho + cap - 1) / cap;
    g.THd = (g.Ho + nty - 1) / nty;
    g.tiles_y = (g.Ho + g.THd - 1) / g.THd;
    g.NI = 1; g.ring = 1;
  }
  g.TH = 2 * g.THd;
  g.LH = g.THd + HL;
  const int lw = g.Wo + CL + CR;
  const bool pow2 = (g.ns & (g.ns - 1)) == 0;
  int lwp = lw;
  if (pow2) { while (lwp % (2 * g.ns) != g.ns) ++lwp; } else if (lwp % 2 == 0) ++lwp;
  g.LWp = lwp;
  g.RH = g.LH;
  int plane = g.NI * g.RH * g.LWp + 4;   // + slack: a half strip reads a fixed number of operand pairs, up to 2 past its last one
  if (plane < 512) plane = 512;
  while (plane % 4 != 2) ++plane;
  g.plane = plane;
  g.TPIX = g.NI * g.TH * W;
  g.TPIXD = g.NI * g.THd * g.Wo;
  int tp = g.TPIX;
  while (tp % 8 != 4) ++tp;
  g.TPIXp = tp;
  g.ntiles = ((N + g.NI - 1) / g.NI) * g.tiles_y;
  g.nslabs = (C + 15) / 16;
  return true;
}

static void cw_workers(CwGeom& g, int per_cu, int max_rows, int nw) {
  if (per_cu < 1) per_cu = 1;
  const int units = g.nslabs * (nw == 4 ? 2 : 1);   // workgroups per worker
  // experiment switch: this launch is one of `share` concurrent ones (the branches of a block on separate streams): 1 / share of the slots
  constexpr int share = 1;
  long want = ((long)num_cus() * per_cu) / units / (share > 1 ? share : 1);
  static const long max_env = getenv("ATOMNAS_DW_MAX_WORKERS") ? atol(getenv("ATOMNAS_DW_MAX_WORKERS")) : 0;   // tests: long tile walks
  if (max_env > 0 && want > max_env) want = max_env;
  if (max_rows > 0 && want > max_rows) want = max_rows;   // every worker owns one partial row
  if (want > g.ntiles) want = g.ntiles;
  if (want < 1) want = 1;
  g.nworkers = (int)want;
}
static unsigned cw_grid(const CwGeom& g, int nw) {
  const unsigned u = (unsigned)g.nworkers * g.nslabs;
  return nw == 4 ? (u + 7) / 8 * 16 : u;   // half-slab workgroups: 8 (slab, worker) units -> 16 blocks, see cw_block
}

static int cw_mode() {
  // bit 0: backward, bit 1: forward (stride 1); bit 2: backward stride 2, bit 3: forward stride 2
  static const int m = getenv("ATOMNAS_DW_CW") ? atoi(getenv("ATOMNAS_DW_CW")) : 7;
  return m;
}
static int cw_nw() {
  constexpr int m = 4;   // waves per workgroup: 8 (whole slab) or 4 (half)
  return m == 8 ? 8 : 4;
}
template <typename T> static size_t cw_lds(const CwGeom& g, int nw) {
  typedef typename Cw<T>::pair_t pair_t;
  return (size_t)nw * g.plane * sizeof(f32x2) + (size_t)nw * g.TPIXp * sizeof(pair_t) + 48 * sizeof(float);
}

}  // namespace atomnas

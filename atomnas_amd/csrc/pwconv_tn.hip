// Weight-gradient GEMMs of the pointwise (1x1) convolutions on gfx950 (split out of pwconv.hip in round 6):
//   gemm_tn  : Out[i,j] += sum_m pro(U)[m,i] * pro(V)[m,j]            (reduction over M = batch x pixels)
// behind atomnas_pw_gemm_tn: dW of ConvBNReLU(inp, hid, 1) (models/mobilenet_base.py:316-320), of the linear projection (:338), of the
// last 1x1 conv (models/mobilenet_supernet.py:148-153) and of the classifier (:160-163).  k_gemm_tn: any storage type (the fp32 parity
// mode); k_gemm_tn2: bf16, 128-row slabs with transposing LDS reads; k_gemm_tn3: bf16, single-stream late-stage shapes on an LDS-DMA ring.
#include "pwconv.h"

namespace atomnas {

// ------------------------------------------------------------------------------------------------ gemm_tn
// Out[i*si + j*sj] += sum_m U[m][i] * V[m][j].  A block owns up to 16*UT_MAX columns of U (all four waves use all of
// them) and 64 columns of V (one 16-column tile per wave) and a contiguous chunk of rows; 32-row slabs of both operands
// are staged (after their prologues) in LDS and read back column-wise as MFMA fragments.
constexpr int UT_MAX = 20;  // 320 channels
constexpr int TN_ROWS = 32;

template <typename T, int UMODE, int VMODE>
__global__ __launch_bounds__(256) void k_gemm_tn(Operand U, int NU, Operand V, int NV, float* __restrict__ out, long si, long sj,
                                                 long M, long rows_per_block, float* __restrict__ ws) {
  using MM = Mma<T>;
  constexpr int E = MM::EPL;
  constexpr int SUB = TN_ROWS / (4 * E);     // MFMA k-steps per 32-row slab (1 for bf16, 8 for fp32)
  constexpr int UP = 16 * UT_MAX + 2;        // LDS pitches (elements); +2 keeps the 8-row groups on distinct banks
  constexpr int VP = 64 + 2;
  __shared__ T s_u[TN_ROWS * UP];
  __shared__ T s_v[TN_ROWS * VP];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;

  const int u0 = blockIdx.z * (16 * UT_MAX);
  const int nu = min(NU - u0, 16 * UT_MAX);       // U columns of this block
  const int ut = (nu + 15) / 16;
  const int v0 = blockIdx.y * 64;
  const int nv = min(NV - v0, 64);
  const long r_beg = (long)blockIdx.x * rows_per_block;
  const long r_end = min(M, r_beg + rows_per_block);

  f32x4 acc[UT_MAX];
#pragma unroll
  for (int t = 0; t < UT_MAX; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ugroups = (ut * 16) / E;  // E-wide channel groups per row in the U slab
  const int vgroups = 64 / E;

  for (long r0 = r_beg; r0 < r_end; r0 += TN_ROWS) {
    // stage U slab [32][ut*16] and V slab [32][64]
    for (int idx = tid; idx < TN_ROWS * ugroups; idx += 256) {
      const int rr = idx / ugroups, cg = (idx % ugroups) * E;
      float v[E];
      load_pro<T, UMODE>(U, r0 + rr, (r0 + rr) < r_end, u0 + cg, NU, v);
#pragma unroll
      for (int e = 0; e < E; ++e) s_u[rr * UP + cg + e] = from_f32<T>(v[e]);
    }
    for (int idx = tid; idx < TN_ROWS * vgroups; idx += 256) {
      const int rr = idx / vgroups, cg = (idx % vgroups) * E;
      float v[E];
      load_pro<T, VMODE>(V, r0 + rr, (r0 + rr) < r_end, v0 + cg, NV, v);
#pragma unroll
      for (int e = 0; e < E; ++e) s_v[rr * VP + cg + e] = from_f32<T>(v[e]);
    }
    __syncthreads();

#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      // B operand: V[m = 4E*sb + E*q + e][col = 16*wave + j]
      float bv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) bv[e] = to_f32(s_v[(4 * E * sb + E * q + e) * VP + 16 * wave + j]);
      const typename MM::frag bf = MM::pack(bv);
#pragma unroll
      for (int t = 0; t < UT_MAX; ++t) {
        if (t < ut) {
          float uv[E];
#pragma unroll
          for (int e = 0; e < E; ++e) uv[e] = to_f32(s_u[(4 * E * sb + E * q + e) * UP + 16 * t + j]);
          acc[t] = MM::mma(MM::pack(uv), bf, acc[t]);
        }
      }
    }
    __syncthreads();
  }

  // D[i = 4q + r][jj = j]: i indexes U columns of tile t, jj the V column 16*wave + j
  const int vc = v0 + 16 * wave + j;
  if (vc < NV) {
#pragma unroll
    for (int t = 0; t < UT_MAX; ++t) {
      if (t < ut) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int uc = u0 + 16 * t + 4 * q + r;
          if (uc < NU) {   // one workgroup per (row chunk, output tile): plain stores, partials summed by reduce_parts
            if (ws) ws[((long)blockIdx.x * NU + uc) * NV + vc] = acc[t][r];
            else out[uc * si + vc * sj] += acc[t][r];
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ gemm_tn, bf16, transposed LDS
// Same contract as k_gemm_tn, restructured for bandwidth: 128-row slabs (4x more bytes in flight per barrier), and the
// operands are written to LDS TRANSPOSED ([column][row], two rows packed per 32-bit store, conflict-free) so that every MFMA
// fragment -- 8 consecutive rows of one column -- is a single ds_read_b128 instead of eight 16-bit reads.
#ifndef TN2_COALESCED
#define TN2_COALESCED 0   // staging loads: 0 = lanes along rows (16 bytes of 64 different lines per instruction, revisited from L1
                          // by the next channel groups), 1 = lanes along channels (whole 128-byte segments).  Measured in situ
                          // (bs 256 step): 11.8 ms vs 13.0 ms per step for all weight-gradient GEMMs -- rows win.
#endif
#ifndef TN2_KS_UNROLL
#define TN2_KS_UNROLL 2
#endif
#ifndef TN_TIMING
#define TN_TIMING 0   // s_memtime phase accounting of k_gemm_tn2 (tools/tnbench2.py, experiment builds only)
#endif
#if TN_TIMING
__device__ unsigned long long g_tn_timing[8];
#define TN_MARK(i)                                                     \
  {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long tn_ = __builtin_readcyclecounter();       \
    tacc[i] += tn_ - tlast;                                            \
    tlast = tn_;                                                       \
    __builtin_amdgcn_sched_barrier(0);                                 \
  }
#else
#define TN_MARK(i)
#endif


// TR (round 4): the operands stay row-major in LDS -- [32-row chunk][16-column tile][32][16] subtiles, filled with ONE 16-byte store per
// (row, 8 channels) piece -- and the k-major MFMA fragments are read with ds_read_b64_tr_b16, gfx950's transposing LDS read (two per
// fragment; semantics checked with tools/probe/trread.hip: lane (c, g) element e <- the 8-byte segment the group's sub-lane 4 e + c / 4
// points at, its element c % 4).  Without it every (row pair, 8 channels) unit is transposed by hand: eight packs and eight 4-byte LDS
// stores, which -- with the loads -- was where the kernel's wave cycles went (profiles/r03_tn_phase_timing.txt).
template <int UMODE, int VMODE, int UTT, int VTT, int ROWS, bool TR>
__global__ __launch_bounds__(256) void k_gemm_tn2(Operand U, int NU, Operand V, int NV, float* __restrict__ out, long si, long sj, long M,
                                                  long rows_per_block, int nchunks, int vt, int uz, int xcd_aware, float* __restrict__ ws) {
  using T = bf16_t;
  using MM = Mma<T>;
  constexpr int VW = 64 * VTT;      // V columns per workgroup: each wave owns VTT tiles of 16 (the U tile is re-read per V tile:
                                    // wider V tiles halve that traffic where U is not narrow)
  constexpr int RP = ROWS + 8;      // transposed row pitch (elements): 16 consecutive columns land on 16 distinct 16-byte slots
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* s_v = reinterpret_cast<T*>(smem_raw);          // [VW][RP]   (TR: [ROWS / 32][VW / 16][32][16])
  T* s_u = s_v + VW * RP;                           // [16*UTT][RP]   (TR: [ROWS / 32][UTT][32][16]; the region is the same size or smaller)
  float* s_cv = reinterpret_cast<float*>(s_u + 16 * UTT * RP);       // [3][VW]      prologue coefficients of the V tile
  float* s_cu = s_cv + 3 * VW;                                       // [3][16*UTT]  ... of the U tile

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  // Workgroup b runs on XCD b % 8.  All column tiles (V tile, U tile) of one row chunk are consecutive workgroups of ONE XCD:
  // they re-read the chunk's narrow operand and share the 128-byte lines that 64-column tiles straddle, so those hit that
  // XCD's L2 instead of going back to HBM once per tile.
  const int ntile = vt * uz;
  int tile; long chunk;
  if (xcd_aware) {
    const int b_xcd = blockIdx.x & 7, b_local = blockIdx.x >> 3;
    tile = b_local % ntile;
    chunk = (long)(b_local / ntile) * 8 + b_xcd;
  } else {
    chunk = blockIdx.x % nchunks;
    tile = blockIdx.x / nchunks;
    if (tile >= ntile) return;
  }
  if (chunk >= nchunks) return;
  const int u0 = (tile / vt) * (16 * UTT);
  const int nu = min(NU - u0, 16 * UTT);
  const int ut = (nu + 15) / 16;
  const int v0 = (tile % vt) * VW;
  const long r_beg = chunk * rows_per_block;
  const long r_end = min(M, r_beg + rows_per_block);

  f32x4 acc[VTT][UTT];
#pragma unroll
  for (int v = 0; v < VTT; ++v)
#pragma unroll
    for (int t = 0; t < UTT; ++t) acc[v][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_coeffs<VMODE>(V, v0, VW, NV, s_cv, tid);
  stage_coeffs<UMODE>(U, u0, 16 * UTT, NU, s_cu, tid);
  __syncthreads();

  const int ugroups = ut * 2;   // 8-channel groups per row
#if TN_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  // Round 3: the staging loads of a slab are issued TOGETHER, one slab ahead, and only then transformed and stored transposed.
  // Before, every staging iteration (two rows x 8 channels per thread) loaded and immediately consumed its own data: 10 memory
  // round trips per 128-row slab (139 us for the 7x7 projection weight gradient, 87 MB of operands).  All loads are unconditional
  // (clamped addresses; validity is applied to the values), so the wait before the transform is a counted one.
  constexpr int NVU = ((ROWS / 2) * (VW / 8) + 255) / 256;        // V work units (row pair x 8 channels) per thread
  constexpr int NUU = ((ROWS / 2) * (2 * UTT) + 255) / 256;       // U work units per thread (at most: ut may be smaller)
  constexpr bool V2 = VMODE == PRO_BNBWD, U2 = UMODE == PRO_BNBWD;
  bf16x8 rva[NVU][2], rvx[V2 ? NVU : 1][2], rua[NUU][2], rux[U2 ? NUU : 1][2];
  const int K8V = (NV + 7) & ~7, K8U = (NU + 7) & ~7;
  // work unit (row pair rp, 8-channel group cg) of a thread.  Slab-major operands: consecutive lanes take the two halves of a slab
  // row and then the next row pair, so that a wave's two loads of a unit cover 2 KB of ONE slab contiguously (16 lines of 128 bytes)
  // -- with the lanes along the rows of one channel group, every load instruction touched 32 lines for 1 KB of payload and the issue of
  // a slab's loads was 42-51 % of the kernel's wave cycles (tools/tntiming.py, r03).  Plain operands: lanes along the rows.
  auto unit = [&](int idx, int groups, bool slab, int& cg, int& rp) {
    if (slab) {
      const int u = idx >> 1;
      rp = u % (ROWS / 2);
      cg = (u / (ROWS / 2)) * 2 + (idx & 1);
    } else {
#if TN2_COALESCED
      cg = idx % groups; rp = idx / groups;
#else
      rp = idx % (ROWS / 2); cg = idx / (ROWS / 2);
#endif
    }
  };
  const bool vslab = V.ss1 != 0, uslab = U.ss1 != 0;
  // per work unit, loop-invariant: the row pair, the byte-free element offset of (row 0, channel group) and the row multiplier of each
  // stream.  Computed per load (layout branch, 64-bit products) the address arithmetic was ~20 instructions and two branches per
  // load, 20 loads per slab.
  const long v_m1 = V.ss1 ? 16 : V.ld1, v_m2 = V.ss2 ? 16 : V.ld2, u_m1 = U.ss1 ? 16 : U.ld1, u_m2 = U.ss2 ? 16 : U.ld2;
  // (row pair, element offsets of (row 0, channel group) in the two streams) of work unit i: needed when the running pointers are set up
  // and in the clamped form of the loads -- not kept in registers across the slab loop
  auto v_unit = [&](int i, int& rp, long& c1, long& c2) {
    int cg;
    const int idx = tid + 256 * i;
    unit(idx < (ROWS / 2) * (VW / 8) ? idx : 0, VW / 8, vslab, cg, rp);
    int k = v0 + cg * 8;
    k = k < NV ? k : K8V - 8;
    c1 = lay_off(0, k, V.ld1, V.ss1);
    c2 = V2 ? lay_off(0, k, V.ld2, V.ss2) : 0;
  };
  auto u_unit = [&](int i, int& rp, long& c1, long& c2) {
    int cg;
    const int idx = tid + 256 * i;
    unit(idx < (ROWS / 2) * ugroups ? idx : 0, ugroups, uslab, cg, rp);
    int k = u0 + cg * 8;
    k = k < NU ? k : K8U - 8;
    c1 = lay_off(0, k, U.ld1, U.ss1);
    c2 = U2 ? lay_off(0, k, U.ld2, U.ss2) : 0;
  };
  // Round 6: the loads of a slab that lies entirely inside the tensor (all but the last one or two) go through RUNNING pointers -- one
  // 64-bit add per load and one per unit and slab -- instead of a clamped row, a 64-bit product and two sums per load (~12 instructions
  // each, 20-30 loads per slab: the kernel issues ~3,000 instructions per slab at two waves per SIMD and is bound by that).  Slabs that
  // touch row M take the clamped form below.
  constexpr bool RUNP = !(U2 && UTT >= 20);   // (twenty two-stream pointers more than that instance has registers for: it spills)
  const T* pv1[NVU]; const T* pv2[V2 ? NVU : 1]; const T* pu1[NUU]; const T* pu2[U2 ? NUU : 1];
#pragma unroll
  for (int i = 0; i < NVU; ++i) {
    int rp; long c1, c2;
    v_unit(i, rp, c1, c2);
    pv1[i] = reinterpret_cast<const T*>(V.p1) + ((r_beg + 2 * rp) * v_m1 + c1);
    if constexpr (V2) pv2[i] = reinterpret_cast<const T*>(V.p2) + ((r_beg + 2 * rp) * v_m2 + c2);
  }
#pragma unroll
  for (int i = 0; i < NUU; ++i) {
    int rp; long c1, c2;
    u_unit(i, rp, c1, c2);
    pu1[i] = reinterpret_cast<const T*>(U.p1) + ((r_beg + 2 * rp) * u_m1 + c1);
    if constexpr (U2) pu2[i] = reinterpret_cast<const T*>(U.p2) + ((r_beg + 2 * rp) * u_m2 + c2);
  }
  long r_run = r_beg;   // the slab the running pointers stand at
  auto issue = [&](long r0) {
    if (RUNP && r0 == r_run && r0 + ROWS <= M) {   // wave-uniform
#pragma unroll
      for (int i = 0; i < NVU; ++i) {
        rva[i][0] = *reinterpret_cast<const bf16x8*>(pv1[i]);
        rva[i][1] = *reinterpret_cast<const bf16x8*>(pv1[i] + v_m1);
        pv1[i] += ROWS * v_m1;
        if constexpr (V2) {
          rvx[i][0] = *reinterpret_cast<const bf16x8*>(pv2[i]);
          rvx[i][1] = *reinterpret_cast<const bf16x8*>(pv2[i] + v_m2);
          pv2[i] += ROWS * v_m2;
        }
      }
#pragma unroll
      for (int i = 0; i < NUU; ++i) {
        rua[i][0] = *reinterpret_cast<const bf16x8*>(pu1[i]);
        rua[i][1] = *reinterpret_cast<const bf16x8*>(pu1[i] + u_m1);
        pu1[i] += ROWS * u_m1;
        if constexpr (U2) {
          rux[i][0] = *reinterpret_cast<const bf16x8*>(pu2[i]);
          rux[i][1] = *reinterpret_cast<const bf16x8*>(pu2[i] + u_m2);
          pu2[i] += ROWS * u_m2;
        }
      }
      r_run = r0 + ROWS;
      return;
    }
#pragma unroll
    for (int i = 0; i < NVU; ++i) {
      int rp; long c1, c2;
      v_unit(i, rp, c1, c2);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        long r = r0 + 2 * rp + h;
        r = r < M ? r : M - 1;
        rva[i][h] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(V.p1) + (r * v_m1 + c1));
        if constexpr (V2) rvx[i][h] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(V.p2) + (r * v_m2 + c2));
      }
    }
#pragma unroll
    for (int i = 0; i < NUU; ++i) {
      int rp; long c1, c2;
      u_unit(i, rp, c1, c2);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        long r = r0 + 2 * rp + h;
        r = r < M ? r : M - 1;
        rua[i][h] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(U.p1) + (r * u_m1 + c1));
        if constexpr (U2) rux[i][h] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(U.p2) + (r * u_m2 + c2));
      }
    }
  };
  // prologue of 8 raw channels of one row (coefficients from LDS; entries of channels >= K are zero), zero where invalid
  auto xform = [&](auto mode_tag, const Operand& o, const bf16x8& ra, const bf16x8& rx, bool guard, bool valid, const float* lc1,
                   const float* lc2, const float* lc3, float (&v)[8]) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)ra[e];
    if constexpr (MODE == PRO_BNRELU) {
      float sc[8], sh[8];
      VecIO<float, 8>::load(lc1, sc);
      VecIO<float, 8>::load(lc2, sh);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
      act_apply_v<8>(v, act_of(o.relu));
    } else if constexpr (MODE == PRO_BNBWD) {
      float a1[8], a2[8], a3[8];
      VecIO<float, 8>::load(lc1, a1);
      VecIO<float, 8>::load(lc2, a2);
      VecIO<float, 8>::load(lc3, a3);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = a1[e] * v[e] + a2[e] * (float)rx[e] + a3[e];
    }
    if (guard) {   // wave-uniform: the slab reaches past the chunk, or a raw operand's tile past its last channel
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = valid ? v[e] : 0.f;
    }
  };
  // the prologues turn the clamped channels past NV / NU into zeros by themselves (their staged coefficients are zero): only a raw
  // operand needs the channel test; rows need theirs in the last slab of the chunk
  const bool v_tail = VMODE == PRO_NONE && v0 + VW > NV, u_tail = UMODE == PRO_NONE && u0 + 16 * UTT > NU;
  auto stage = [&](long r0) {
    const bool row_guard = r0 + ROWS > r_end;
#pragma unroll
    for (int i = 0; i < NVU; ++i) {
      int cg, rp;
      const int idx = tid + 256 * i;
      const bool active = idx < (ROWS / 2) * (VW / 8);
      unit(active ? idx : 0, VW / 8, vslab, cg, rp);
      float a[8], bb[8];
      const float* lc = s_cv + cg * 8;
      const bool kv = v0 + cg * 8 < NV;
      xform(std::integral_constant<int, VMODE>{}, V, rva[i][0], rvx[V2 ? i : 0][0], row_guard || v_tail, kv && (r0 + 2 * rp) < r_end, lc, lc + VW, lc + 2 * VW, a);
      xform(std::integral_constant<int, VMODE>{}, V, rva[i][1], rvx[V2 ? i : 0][1], row_guard || v_tail, kv && (r0 + 2 * rp + 1) < r_end, lc, lc + VW, lc + 2 * VW, bb);
      const int rot = TN2_COALESCED ? 2 * (cg >> 1) : 0;
      if (active) {
        if constexpr (TR) {
          // row-major subtiles [2 rp / 32][cg / 2][32][16]: one 16-byte store per row
          T* d = s_v + (((2 * rp) >> 5) * (VW / 16) + (cg >> 1)) * 512 + ((2 * rp) & 31) * 16 + (cg & 1) * 8;
          *reinterpret_cast<bf16x8*>(d) = MM::pack(a);
          *reinterpret_cast<bf16x8*>(d + 16) = MM::pack(bb);
        } else {
#pragma unroll
          for (int ii = 0; ii < 8; ++ii) {
            const int e = (ii + rot) & 7;
            bf16x2 pk;
            pk[0] = (bf16_t)a[e]; pk[1] = (bf16_t)bb[e];
            *reinterpret_cast<bf16x2*>(&s_v[(cg * 8 + e) * RP + 2 * rp]) = pk;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NUU; ++i) {
      int cg, rp;
      const int idx = tid + 256 * i;
      const bool active = idx < (ROWS / 2) * ugroups;
      unit(active ? idx : 0, ugroups, uslab, cg, rp);
      float a[8], bb[8];
      const float* lc = s_cu + cg * 8;
      const bool kv = u0 + cg * 8 < NU;
      xform(std::integral_constant<int, UMODE>{}, U, rua[i][0], rux[U2 ? i : 0][0], row_guard || u_tail, kv && (r0 + 2 * rp) < r_end, lc, lc + 16 * UTT, lc + 32 * UTT, a);
      xform(std::integral_constant<int, UMODE>{}, U, rua[i][1], rux[U2 ? i : 0][1], row_guard || u_tail, kv && (r0 + 2 * rp + 1) < r_end, lc, lc + 16 * UTT, lc + 32 * UTT, bb);
      const int rot = TN2_COALESCED ? 2 * (cg >> 1) : 0;
      if (active) {
        if constexpr (TR) {
          T* d = s_u + (((2 * rp) >> 5) * UTT + (cg >> 1)) * 512 + ((2 * rp) & 31) * 16 + (cg & 1) * 8;
          *reinterpret_cast<bf16x8*>(d) = MM::pack(a);
          *reinterpret_cast<bf16x8*>(d + 16) = MM::pack(bb);
        } else {
#pragma unroll
          for (int ii = 0; ii < 8; ++ii) {
            const int e = (ii + rot) & 7;
            bf16x2 pk;
            pk[0] = (bf16_t)a[e]; pk[1] = (bf16_t)bb[e];
            *reinterpret_cast<bf16x2*>(&s_u[(cg * 8 + e) * RP + 2 * rp]) = pk;
          }
        }
      }
    }
  };
  // TR: fragment of subtile `st` (a [32][16] row-major block, 1 KB) -- k = 8 q + e along the rows, column j: two transposing reads
  // (rows 8 q .. 8 q + 3 and 8 q + 4 .. 8 q + 7); every lane passes the address of its 8-byte segment: row 8 q + j / 4, columns 4 (j % 4) ..
  const unsigned tr_lane = (unsigned)(((8 * q + (j >> 2)) * 16 + 4 * (j & 3)) * 2);
  auto tr_frag = [&](const T* sub) {
    const unsigned a0 = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)sub) + tr_lane;
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:128" : "=v"(hi) : "v"(a0) : "memory");
    return TrFrag{lo, hi};
  };
  issue(r_beg);
  for (long r0 = r_beg; r0 < r_end; r0 += ROWS) {
    TN_MARK(5)
    stage(r0);
    TN_MARK(0)
    issue(r0 + ROWS);   // the next slab flies during the MFMA phase (past the chunk: clamped rows, values never used)
    TN_MARK(1)
    __syncthreads();
    TN_MARK(2)
#pragma unroll TN2_KS_UNROLL   // full unrolling hoists all fragment reads: 162 VGPRs for 6 accumulator tiles, 2 waves per SIMD
    for (int ks = 0; ks < ROWS / 32; ++ks) {
      if constexpr (TR) {
        // all fragment reads of the k-step are issued, ONE wait (the asm reads are invisible to the compiler's counters), then the MFMAs
        TrFrag bfp[VTT], afp[UTT];
#pragma unroll
        for (int v = 0; v < VTT; ++v) bfp[v] = tr_frag(s_v + (ks * (VW / 16) + 4 * v + wave) * 512);
#pragma unroll
        for (int t = 0; t < UTT; ++t)
          if (t < ut) afp[t] = tr_frag(s_u + (ks * UTT + t) * 512);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 bf[VTT];
#pragma unroll
        for (int v = 0; v < VTT; ++v) bf[v] = __builtin_shufflevector(bfp[v].lo, bfp[v].hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int t = 0; t < UTT; ++t) {
          if (t < ut) {
            const bf16x8 af = __builtin_shufflevector(afp[t].lo, afp[t].hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int v = 0; v < VTT; ++v) acc[v][t] = MM::mma(af, bf[v], acc[v][t]);
          }
        }
      } else {
        bf16x8 bf[VTT];
#pragma unroll
        for (int v = 0; v < VTT; ++v) bf[v] = *reinterpret_cast<const bf16x8*>(&s_v[(64 * v + 16 * wave + j) * RP + 32 * ks + 8 * q]);
#pragma unroll
        for (int t = 0; t < UTT; ++t) {
          if (t < ut) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(&s_u[(16 * t + j) * RP + 32 * ks + 8 * q]);
#pragma unroll
            for (int v = 0; v < VTT; ++v) acc[v][t] = MM::mma(af, bf[v], acc[v][t]);
          }
        }
      }
    }
    TN_MARK(3)
    __syncthreads();
    TN_MARK(4)
  }
#if TN_TIMING
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&g_tn_timing[i], tacc[i]);
    atomicAdd(&g_tn_timing[7], (unsigned long long)((r_end - r_beg + ROWS - 1) / ROWS));
  }
#endif

#pragma unroll
  for (int v = 0; v < VTT; ++v) {
    const int vc = v0 + 64 * v + 16 * wave + j;
    if (vc < NV) {
#pragma unroll
      for (int t = 0; t < UTT; ++t) {
        if (t < ut) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int uc = u0 + 16 * t + 4 * q + r;
            if (uc < NU) {   // one workgroup per (row chunk, output tile): plain stores, partials summed by reduce_parts
              if (ws) ws[(chunk * NU + uc) * NV + vc] = acc[v][t][r];
              else out[uc * si + vc * sj] += acc[v][t][r];
            }
          }
        }
      }
    }
  }
}

template <int UTT, int VTT, int ROWS>
static int launch_tn2_cfg(int umode, const Operand& U, int NU, int vmode, const Operand& V, int NV, float* out, long si, long sj, long M,
                          float* ws, long ws_floats, hipStream_t st) {
  const int vt = (NV + 64 * VTT - 1) / (64 * VTT), uz = (NU + 16 * UTT - 1) / (16 * UTT);
  const size_t lds = (size_t)(64 * VTT + 16 * UTT) * (ROWS + 8) * sizeof(bf16_t) + (size_t)3 * (64 * VTT + 16 * UTT) * sizeof(float);
  // row chunks so that the grid is one round of resident workgroups (at least two slabs per workgroup)
  constexpr int xcd_env = 1;
  constexpr int tr_on = 1;   // experiment switch: transposing LDS reads (round 4)
#define TN2_CASE(UM, VM)                                                                                                      \
  {                                                                                                                           \
    auto kern = tr_on ? k_gemm_tn2<UM, VM, UTT, VTT, ROWS, true> : k_gemm_tn2<UM, VM, UTT, VTT, ROWS, false>;                 \
    const long resident = (long)num_cus() * resident_per_cu(kern, 256, lds);                                                  \
    long chunks = resident / ((long)vt * uz);                                                                                 \
    if (chunks > M / (2 * ROWS)) chunks = M / (2 * ROWS);                                                                     \
    if (chunks > max_chunks) chunks = max_chunks;   /* every row chunk owns one partial output in the workspace */           \
    int xcd = xcd_env;                                                                                                        \
    if (chunks < 8) xcd = 0; /* fewer chunks than XCDs: plain order */                                                        \
    if (xcd) chunks = chunks / 8 * 8; /* equal work per XCD */                                                                \
    if (chunks < 1) chunks = 1;                                                                                               \
    const long rows = (M + chunks - 1) / chunks;                                                                              \
    chunks = (M + rows - 1) / rows;                                                                                           \
    nparts = chunks;                                                                                                          \
    dim3 grid((unsigned)((xcd ? (chunks + 7) / 8 * 8 : chunks) * vt * uz)), block(256);                                       \
    hipLaunchKernelGGL(kern, grid, block, lds, st, U, NU, V, NV, out, si, sj, M, rows, (int)chunks, vt, uz, xcd,             \
                       chunks > 1 ? ws : nullptr);                                                                            \
  }
  const long max_chunks = (ws && (long)NU * NV > 0) ? ws_floats / ((long)NU * NV) : 1;   // < 2: single chunk, direct accumulation
  long nparts = 1;
  if (umode == PRO_NONE && vmode == PRO_NONE) TN2_CASE(PRO_NONE, PRO_NONE)
  else if (umode == PRO_NONE && vmode == PRO_BNBWD) TN2_CASE(PRO_NONE, PRO_BNBWD)
  else if (umode == PRO_BNBWD && vmode == PRO_BNRELU) TN2_CASE(PRO_BNBWD, PRO_BNRELU)
  else if (umode == PRO_BNRELU && vmode == PRO_BNBWD) TN2_CASE(PRO_BNRELU, PRO_BNBWD)
  else if (umode == PRO_BNBWD && vmode == PRO_NONE) TN2_CASE(PRO_BNBWD, PRO_NONE)
  else if (umode == PRO_NONE && vmode == PRO_BNRELU) TN2_CASE(PRO_NONE, PRO_BNRELU)   /* U = atomnas_bnbwd_apply's output */
  else { set_error("gemm_tn: unsupported prologue pair (%d,%d)", umode, vmode); return 1; }
#undef TN2_CASE
  if (int rc = check_launch("gemm_tn2")) return rc;
  if (nparts > 1) return reduce_parts(ws, (long)NU * NV, (int)nparts, (long)NU * NV, out, NV, si, sj, st);
  return 0;
}

// V tile width.  The U slab is re-staged (transposed) for every V tile, so with a wide U (many accumulator tiles) a 128-column V tile
// halves that work at the price of LDS / registers.  Measured in situ per shape (bs 256 step, after the early-stage weight gradients
// moved into the fused backward kernels; profiles/r02_bs256_per_shape_timing.txt): NU 40..96 (4 / 6 accumulator tiles): 128 columns
// on 64-row slabs -15..-30 %; NU 320 (two U tiles of 160): 128 columns on 128-row slabs -20 %; NU 192 (12 tiles): 64 columns stay
// best (+20..35 % otherwise).  ATOMNAS_TN_WIDE = 0 / 1 / 2 forces one form (A/B).
template <int UTT>
static int launch_tn2_ut(int umode, const Operand& U, int NU, int vmode, const Operand& V, int NV, float* out, long si, long sj, long M,
                         float* ws, long ws_floats, hipStream_t st) {
  constexpr int wide_env = -1;
  if constexpr (UTT >= 4 && UTT <= 12) {
    // r03, after the staging rewrite (same-call A/B over the step's shapes, ATOMNAS_TN_WIDE=0/1/2: 4.36 / 3.46 / 3.88 ms): 128-column V
    // tiles on 64-row slabs win for every U width (r02 had 64-column tiles for 12 U tiles and 128-row slabs for 10)
    const int wide = wide_env >= 0 ? wide_env : 1;
    if (wide == 1 && NV >= 256) return launch_tn2_cfg<UTT, 2, 64>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
    if (wide == 2 && NV >= 256) return launch_tn2_cfg<UTT, 2, 128>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  }
  return launch_tn2_cfg<UTT, 1, 128>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
}

// ------------------------------------------------------------------------------------------------ gemm_tn, LDS-DMA form (round 4)
// Out[i][j] (+)= sum_m U[m][i] * pro(V)[m][j] for the single-stream weight gradients of the late stages (U: plain, no prologue -- the
// block input x or the differentiated BatchNorm output dP; V: a hidden tensor with no prologue or the BNRELU one).  k_gemm_tn2 stages both
// operands through registers: a slab's loads are issued one slab ahead, so every slab waits out what is left of an HBM round trip, and
// the staging itself (loads, prologue, LDS stores) was 36-51 % of its wave cycles; its matrix pipe is 6 % busy.  Here:
//   * global_load_lds_dwordx4 copies 32-row x 16-column subtiles ([32][16] row-major, 1 KB = one wave instruction, lane-linear) straight
//     into a ring of TN3_DEPTH stage buffers: no staging registers, no LDS store instructions, TN3_DEPTH - 1 stages in flight behind
//     the MFMAs, waited for with a COUNTED vmcnt (every wave issues the same number of copies per stage);
//   * the k-major MFMA fragments come out of those row-major subtiles with ds_read_b64_tr_b16;
//   * V's prologue (scale / shift / activation, per COLUMN, i.e. per lane of a B fragment) is applied to the fragment registers;
//   * rows beyond M are cut off in U's fragments (the copies of such rows read row M - 1: finite garbage times zero).
// A workgroup owns every U column of a U tile (<= 192) and 128 V columns (wave w: V tiles w and w + 4), and a row chunk; partial
// outputs per row chunk go to the workspace and are summed in chunk order (reduce_parts), as in k_gemm_tn2.
// (Tried and removed: copying a dense U's 32 stage rows as ONE contiguous run of 64 NU bytes, read back with the row pitch NU --
//  whole 128-byte lines per copy instead of 32-byte pieces of 32 lines.  1.355 against 1.235 ms over the step's 16 launches.)
template <int UTT, bool VPRO, int TN3_DEPTH>
__global__ __launch_bounds__(256, 2) void k_gemm_tn3(Operand U, int NU, Operand V, int NV, float* __restrict__ out, long si, long sj, long M,
                                                  long rows_per_block, int nchunks, int vt, int uz, float* __restrict__ ws) {
  using T = bf16_t;
  using MM = Mma<T>;
  constexpr int NSUB = 8 + UTT;              // subtiles of a stage: 8 V tiles, UTT U tiles
  constexpr int DPS = (NSUB + 3) / 4;        // copies per wave and stage
  constexpr int STAGE = NSUB * 512;          // elements per stage buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* s_st = reinterpret_cast<T*>(smem_raw);  // [TN3_DEPTH][NSUB][32][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int ntile = vt * uz;
  // all column tiles of one row chunk are consecutive workgroups of ONE XCD (block b runs on XCD b % 8): they share the chunk's U rows in L2
  const int b_xcd = blockIdx.x & 7, b_local = blockIdx.x >> 3;
  const int tile = b_local % ntile;
  const long chunk = (long)(b_local / ntile) * 8 + b_xcd;
  if (chunk >= nchunks) return;
  const int u0 = (tile / vt) * (16 * UTT);
  const int nu = min(NU - u0, 16 * UTT);
  const int ut = (nu + 15) / 16;
  const int v0 = (tile % vt) * 128;
  const long r_beg = chunk * rows_per_block;
  const long r_end = min(M, r_beg + rows_per_block);
  const int nstages = (int)((r_end - r_beg + 31) / 32);

  // per-lane source offsets (elements) of the subtiles this wave copies: lane -> (row lane / 2, 8-column half lane % 2)
  const int r_l = lane >> 1, h_l = lane & 1;
  long src_off[DPS], src_mul[DPS];      // element offset of (row 0) and the row multiplier
  const T* src_base[DPS];
  unsigned dst_off[DPS];                // byte offset of the subtile inside a stage buffer
#pragma unroll
  for (int i = 0; i < DPS; ++i) {
    int sub = wave + 4 * i;
    sub = sub < NSUB ? sub : NSUB - 1;   // surplus slot: the last subtile once more (identical bytes to the same place)
    dst_off[i] = (unsigned)sub * 1024u;
    if (sub < 8) {   // V tile `sub` of the workgroup's 128 columns
      int c = v0 + 16 * sub + 8 * h_l;
      c = c < ((NV + 7) & ~7) ? c : 0;   // a tile beyond NV: any valid column (its results are not stored)
      src_base[i] = reinterpret_cast<const T*>(V.p1);
      src_off[i] = lay_off(0, c, V.ld1, V.ss1);
      src_mul[i] = V.ss1 ? 16 : V.ld1;
    } else {         // U tile sub - 8
      int c = u0 + 16 * (sub - 8) + 8 * h_l;
      c = c < ((NU + 7) & ~7) ? c : 0;
      src_base[i] = reinterpret_cast<const T*>(U.p1);
      src_off[i] = lay_off(0, c, U.ld1, U.ss1);
      src_mul[i] = U.ss1 ? 16 : U.ld1;
    }
  }
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)s_st);
  auto copy_stage = [&](int s) {   // stage s of this chunk -> ring slot s % TN3_DEPTH; rows clamped to M - 1
    const int sc = s < nstages ? s : nstages - 1;   // past the chunk: re-copy its last stage (keeps the copy count per stage constant)
    long row = r_beg + 32L * sc + r_l;
    row = row < M ? row : M - 1;
    const unsigned slot = lds0 + (unsigned)(s % TN3_DEPTH) * (unsigned)(STAGE * 2);
#pragma unroll
    for (int i = 0; i < DPS; ++i) {
      const T* g = src_base[i] + (row * src_mul[i] + src_off[i]);
      const unsigned d = __builtin_amdgcn_readfirstlane(slot + dst_off[i]);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(g), "s"(d) : "memory");
    }
    ATOMNAS_RING_STAGE_END();
  };

  // V prologue coefficients of this lane's two columns (B fragment: lane j <-> column)
  float vsc[2] = {1.f, 1.f}, vsh[2] = {0.f, 0.f};
  if constexpr (VPRO) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int c = v0 + 64 * v + 16 * wave + j;
      if (c < NV) { vsc[v] = V.c1[c]; vsh[v] = V.c2[c]; }
    }
  }
  const Act vact = act_of(V.relu);

  f32x4 acc[2][UTT];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int t = 0; t < UTT; ++t) acc[v][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned tr_lane = (unsigned)(((8 * q + (j >> 2)) * 16 + 4 * (j & 3)) * 2);
  auto tr_frag = [&](unsigned sub_addr) {
    const unsigned a0 = sub_addr + tr_lane;
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:128" : "=v"(hi) : "v"(a0) : "memory");
    return TrFrag{lo, hi};
  };

  // prologue: TN3_DEPTH - 1 stages in flight
#pragma unroll
  for (int s = 0; s < TN3_DEPTH - 1; ++s) copy_stage(s);
  for (int s = 0; s < nstages; ++s) {
    // stage s has landed when at most (TN3_DEPTH - 2) later stages' copies of this wave are outstanding; the barrier extends that to
    // the copies of the other waves
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TN3_DEPTH - 2) * DPS) : "memory");
    __syncthreads();
    copy_stage(s + TN3_DEPTH - 1);   // into the slot stage s - 1 was read from (every wave is past it: the barrier above)
    const unsigned slot = lds0 + (unsigned)(s % TN3_DEPTH) * (unsigned)(STAGE * 2);
    TrFrag bfp[2], afp[UTT];
#pragma unroll
    for (int v = 0; v < 2; ++v) bfp[v] = tr_frag(slot + (unsigned)(4 * v + wave) * 1024u);
#pragma unroll
    for (int t = 0; t < UTT; ++t)
      if (t < ut) afp[t] = tr_frag(slot + (unsigned)(8 + t) * 1024u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 bf[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      bf[v] = __builtin_shufflevector(bfp[v].lo, bfp[v].hi, 0, 1, 2, 3, 4, 5, 6, 7);
      if constexpr (VPRO) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)bf[v][e] * vsc[v] + vsh[v];
        act_apply_v<8>(x, vact);
        bf[v] = MM::pack(x);
      }
    }
    const long rows_left = r_end - (r_beg + 32L * s);   // < 32 only in the chunk's last stage
#pragma unroll
    for (int t = 0; t < UTT; ++t) {
      if (t < ut) {
        bf16x8 af = __builtin_shufflevector(afp[t].lo, afp[t].hi, 0, 1, 2, 3, 4, 5, 6, 7);
        if (rows_left < 32) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (8 * q + e >= rows_left) af[e] = (bf16_t)0.f;
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[v][t] = MM::mma(af, bf[v], acc[v][t]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus copies of the last stages must not outlive the workgroup's LDS

#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int vc = v0 + 64 * v + 16 * wave + j;
    if (vc < NV) {
#pragma unroll
      for (int t = 0; t < UTT; ++t) {
        if (t < ut) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int uc = u0 + 16 * t + 4 * q + r;
            if (uc < NU) {
              if (ws) ws[(chunk * NU + uc) * NV + vc] = acc[v][t][r];
              else out[uc * si + vc * sj] += acc[v][t][r];
            }
          }
        }
      }
    }
  }
}

// -1: not one of this kernel's cases
template <int UTT>
static int launch_tn3_cfg(const Operand& U, int NU, int vmode, const Operand& V, int NV, float* out, long si, long sj, long M, float* ws,
                          long ws_floats, hipStream_t st) {
  const int vt = (NV + 127) / 128, uz = (NU + 16 * UTT - 1) / (16 * UTT);
  // ring depth: as many stages in flight as leave room for two workgroups per CU (ATOMNAS_TN3_DEPTH: experiment switch)
  constexpr int depth_env = 0;
  const int depth = depth_env ? depth_env : ((size_t)4 * (8 + UTT) * 1024 * 2 + 4096 <= max_lds_bytes() ? 4 : 3);
  const size_t lds = (size_t)depth * (8 + UTT) * 1024;
  if (lds > max_lds_bytes()) return -1;
  constexpr bool dbg = false;
  const long max_chunks = (ws && (long)NU * NV > 0) ? ws_floats / ((long)NU * NV) : 1;
  long nparts = 1;
#define TN3_CASE(VP)                                                                                                          \
  {                                                                                                                           \
    auto kern = depth == 4 ? k_gemm_tn3<UTT, VP, 4> : (depth == 3 ? k_gemm_tn3<UTT, VP, 3> : k_gemm_tn3<UTT, VP, 2>);          \
    const long resident = (long)num_cus() * resident_per_cu(kern, 256, lds);                                                  \
    if (dbg) fprintf(stderr, "tn3: M %ld NU %d NV %d UTT %d depth %d lds %zu per_cu %ld\n", M, NU, NV, UTT, depth, lds, resident / num_cus()); \
    long chunks = resident / ((long)vt * uz);                                                                                 \
    if (chunks > M / 128) chunks = M / 128;   /* at least four stages per workgroup */                                        \
    if (chunks > max_chunks) chunks = max_chunks;                                                                             \
    chunks = chunks / 8 * 8;   /* equal work per XCD */                                                                       \
    if (chunks < 8) return -1;                                                                                                \
    long rows = (M + chunks - 1) / chunks;                                                                                    \
    rows = (rows + 31) / 32 * 32;   /* whole stages: only the tensor's last chunk has a ragged one */                         \
    chunks = (M + rows - 1) / rows;                                                                                           \
    nparts = chunks;                                                                                                          \
    dim3 grid((unsigned)((chunks + 7) / 8 * 8 * vt * uz)), block(256);                                                        \
    hipLaunchKernelGGL(kern, grid, block, lds, st, U, NU, V, NV, out, si, sj, M, rows, (int)chunks, vt, uz, ws);              \
  }
  if (vmode == PRO_BNRELU) TN3_CASE(true) else TN3_CASE(false)
#undef TN3_CASE
  if (int rc = check_launch("gemm_tn3")) return rc;
  return reduce_parts(ws, (long)NU * NV, (int)nparts, (long)NU * NV, out, NV, si, sj, st);
}

static int launch_tn3(int umode, const Operand& U, int NU, int vmode, const Operand& V, int NV, float* out, long si, long sj, long M,
                      float* ws, long ws_floats, hipStream_t st) {
  static const int on = getenv("ATOMNAS_TN_DMA") ? atoi(getenv("ATOMNAS_TN_DMA")) : 1;   // experiment switch
  // single-stream weight gradients of wide hidden tensors: U without prologue, V none / BNRELU, a workspace for >= 8 row chunks
  if (!on || umode != PRO_NONE || (vmode != PRO_NONE && vmode != PRO_BNRELU) || !ws || NV < 256 || NU < 32 || M < 1024) return -1;
  if (vmode == PRO_BNRELU && !(V.c1 && V.c2)) return -1;
  const int ut = (NU + 15) / 16;
  if (ut <= 4) return launch_tn3_cfg<4>(U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  if (ut <= 6) return launch_tn3_cfg<6>(U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  if (ut <= 12) return launch_tn3_cfg<12>(U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  if (ut <= 20) return launch_tn3_cfg<10>(U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);   // two U tiles of <= 160 columns
  return -1;
}

static int launch_tn2(int umode, const Operand& U, int NU, int vmode, const Operand& V, int NV, float* out, long si, long sj, long M,
                      float* ws, long ws_floats, hipStream_t st) {
  {
    const int rc = launch_tn3(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
    if (rc >= 0) return rc;
  }
  // accumulator tiles per wave (= U tiles of 16 columns): fewer tiles -> fewer AGPRs -> more waves per SIMD
  const int ut = (NU + 15) / 16;
  if (ut <= 2) return launch_tn2_ut<2>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  if (ut <= 4) return launch_tn2_ut<4>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  if (ut <= 6) return launch_tn2_ut<6>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  if (ut <= 12) return launch_tn2_ut<12>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  if (ut <= 20) return launch_tn2_ut<10>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);   // two U tiles of <= 160 columns
  return launch_tn2_ut<20>(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
}

template <typename T>
static int launch_tn(int umode, const Operand& U, int NU, int vmode, const Operand& V, int NV, float* out, long si, long sj, long M,
                     float* ws, long ws_floats, hipStream_t st) {
  if constexpr (sizeof(T) == 2) return launch_tn2(umode, U, NU, vmode, V, NV, out, si, sj, M, ws, ws_floats, st);
  const int vt = (NV + 63) / 64, uz = (NU + 16 * UT_MAX - 1) / (16 * UT_MAX);
  // enough row chunks to fill the chip, but at least 8 slabs of 32 rows per block
  long chunks = (1024 + (long)vt * uz - 1) / ((long)vt * uz);
  const long max_chunks = ws ? ws_floats / ((long)NU * NV) : 1;   // every row chunk owns one partial output in the workspace
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  long rows = (M + chunks - 1) / chunks;
  if (rows < 8 * TN_ROWS) rows = 8 * TN_ROWS;
  rows = (rows + TN_ROWS - 1) / TN_ROWS * TN_ROWS;
  chunks = (M + rows - 1) / rows;
  dim3 grid((unsigned)chunks, vt, uz), block(256);
#define TN_CASE(UM, VM) \
  hipLaunchKernelGGL((k_gemm_tn<T, UM, VM>), grid, block, 0, st, U, NU, V, NV, out, si, sj, M, rows, chunks > 1 ? ws : nullptr)
  if (umode == PRO_NONE && vmode == PRO_NONE) TN_CASE(PRO_NONE, PRO_NONE);
  else if (umode == PRO_NONE && vmode == PRO_BNBWD) TN_CASE(PRO_NONE, PRO_BNBWD);
  else if (umode == PRO_BNBWD && vmode == PRO_BNRELU) TN_CASE(PRO_BNBWD, PRO_BNRELU);
  else if (umode == PRO_BNRELU && vmode == PRO_BNBWD) TN_CASE(PRO_BNRELU, PRO_BNBWD);
  else if (umode == PRO_BNBWD && vmode == PRO_NONE) TN_CASE(PRO_BNBWD, PRO_NONE);
  else if (umode == PRO_NONE && vmode == PRO_BNRELU) TN_CASE(PRO_NONE, PRO_BNRELU);
  else { set_error("gemm_tn: unsupported prologue pair (%d,%d)", umode, vmode); return 1; }
#undef TN_CASE
  if (int rc = check_launch("gemm_tn")) return rc;
  if (chunks > 1) return reduce_parts(ws, (long)NU * NV, (int)chunks, (long)NU * NV, out, NV, si, sj, st);
  return 0;
}

}  // namespace atomnas

#if TN_TIMING
extern "C" int atomnas_debug_tn_timing(unsigned long long* out8, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(atomnas::g_tn_timing), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(atomnas::g_tn_timing), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif

using namespace atomnas;

// out[i*si + j*sj] += sum_m prologue(U)[m,i] * prologue(V)[m,j]   (fp32 accumulation into `out`, caller zeroes it).
// ws: caller-owned scratch of ws_floats floats for the per-row-chunk partial outputs (summed in chunk order, no atomics:
// bit-reproducible); with ws == NULL or room for fewer than two partials the reduction over M runs in a single workgroup per tile.
extern "C" int atomnas_pw_gemm_tn(int u_mode, const void* u, int ldu, long u_ss, const void* u2, int ldu2, long u2_ss, const float* uc1,
                                  const float* uc2, const float* uc3, int u_relu, int NU, int v_mode, const void* v, int ldv, long v_ss,
                                  const void* v2, int ldv2, long v2_ss, const float* vc1, const float* vc2, const float* vc3, int v_relu, int NV, float* out,
                                  long si, long sj, long M, float* ws, long ws_floats, int dtype, void* stream) {
  ATOMNAS_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "pw_gemm_tn: bad dtype %d", dtype);
  ATOMNAS_REQUIRE(M > 0 && NU > 0 && NV > 0 && out, "pw_gemm_tn: empty shape");
  Operand U{u, ldu, u2, ldu2, u_ss, u2_ss, uc1, uc2, uc3, u_relu};
  Operand V{v, ldv, v2, ldv2, v_ss, v2_ss, vc1, vc2, vc3, v_relu};
  ATOMNAS_REQUIRE((u_ss == 0 || u_ss >= M * 16) && (v_ss == 0 || v_ss >= M * 16), "pw_gemm_tn: slab stride smaller than M*16");
  if (check_operand("pw_gemm_tn(U)", U, u_mode, NU)) return 1;
  if (check_operand("pw_gemm_tn(V)", V, v_mode, NV)) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32) return launch_tn<float>(u_mode, U, NU, v_mode, V, NV, out, si, sj, M, ws, ws_floats, st);
  return launch_tn<bf16_t>(u_mode, U, NU, v_mode, V, NV, out, si, sj, M, ws, ws_floats, st);
}


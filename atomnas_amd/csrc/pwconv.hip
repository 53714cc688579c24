// Pointwise (1x1) convolutions of the atomic block as MFMA GEMMs on gfx950.
//
// Replaces the ATen conv2d/1x1 calls behind ConvBNReLU(inp, hid, 1) (models/mobilenet_base.py:316-320), the linear
// projection nn.Conv2d(hid, oup, 1) (:338), the last 1x1 conv (models/mobilenet_supernet.py:148-153) and the classifier
// (:160-163), forward and backward.  All three branches of a block are concatenated along the hidden dimension
// (sum_i W_i h_i == [W_1 W_2 W_3][h_1;h_2;h_3], mobilenet_base.py:378), so a block is one expand and one project GEMM.
//
//   gemm_nt  : C[M,N] = epi( pro(A)[M,K] * Wp[N,K]^T )               (forward and input-gradient form)
//   gemm_tn  : Out[i,j] += sum_m pro(U)[m,i] * pro(V)[m,j]            (weight-gradient form, reduction over M)
//
// These GEMMs are extremely skinny (min(K,N) is 16..320 while M = batch*H*W is up to 3.2e6): they are HBM-bound, so the
// design goal is to stream A / C exactly once with 16-byte accesses and hide everything else behind it:
//   * prologue fusions on the A side: BatchNorm-apply + ReLU (forward) or the BatchNorm-backward affine
//     dX = c1*g + c2*x + c3 of two streams, so normalised / differentiated activations never round-trip through HBM;
//   * epilogue fusions: residual add, ReLU mask of the producer, and the per-channel sums the next BatchNorm
//     (forward: sum c, sum c^2; backward: sum g, sum g*x) needs;
//   * mfma_f32_16x16x32_bf16 with the weight matrix as the MFMA "A" operand: the accumulator layout then gives every lane
//     16 consecutive output channels of one pixel, i.e. whole 32-byte runs, with no LDS transpose.
// fp32 storage uses mfma_f32_16x16x4f32 through the same code (exact f32; for parity tests, not for speed).
#include "pwconv.h"

namespace atomnas {

// ------------------------------------------------------------------------------------------------ gemm_nt
// One wave owns 16 rows of A per step and produces 64 output channels at a time with 4 MFMA tiles whose weight rows
// are interleaved (row i of tile t is channel nc + 16*(i>>2) + 4*t + (i&3)) so that lane (q = lane>>4, j = lane&15)
// ends up with channels nc+16q .. nc+16q+15 of pixel m0+j.
template <typename T, int MODE, int NCG>
__global__ __launch_bounds__(256) void k_gemm_nt(Operand A, const T* __restrict__ Wp, int ldw, Epilogue ep, long M, int N, int K,
                                                 int Kpad) {
  using MM = Mma<T>;
  constexpr int E = MM::EPL;
  constexpr int KS = 4 * E;
  constexpr int SWD = 64 * NCG;
  extern __shared__ float s_stat[];  // [4 waves][2][SWD] when statistics are taken

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int wrow = 16 * (j >> 2) + (j & 3);  // weight row of this lane inside a 64-channel chunk, before the +4*t
  const bool do_stats = ep.stats != nullptr && ep.stat_mode != STAT_NONE;
  float* sw = s_stat + wave * 2 * SWD;

  if (do_stats) {
    for (int i = tid; i < 8 * SWD; i += 256) s_stat[i] = 0.f;
    __syncthreads();
  }

  // A workgroup owns ONE group of NCG 64-channel chunks (grp) and the row slot rs of R: its waves walk the 16-row tiles
  // rs*4 + wave, + 4R, ...  (small-M problems -- classifier, 7x7 maps -- still fill the chip through the groups).  The fixed
  // channel group is what lets the statistics live in small wave-private LDS rows and leave as one partial row per slot.
  const long mtiles = (M + 15) / 16;
  const int ngroups = (N + SWD - 1) / SWD;
  const int grp = blockIdx.x % ngroups, rs = blockIdx.x / ngroups, R = gridDim.x / ngroups;
  const int nc0 = grp * SWD;
  for (long mt = (long)rs * 4 + wave; mt < mtiles; mt += (long)R * 4) {
    const long m0 = mt * 16;
    const long row = m0 + j;
    const bool rowvalid = row < M;
    {
      f32x4 acc[NCG][4];
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
      for (int k0 = 0; k0 < Kpad; k0 += KS) {
        const int k = k0 + q * E;
        float av[E];
        load_pro<T, MODE>(A, row, rowvalid, k, K, av);
        const typename MM::frag af = MM::pack(av);
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
          const int nc = nc0 + 64 * g;
          if (nc < N) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const typename MM::frag wf = MM::raw(Wp + (long)(nc + wrow + 4 * t) * ldw + k);
              acc[g][t] = MM::mma(wf, af, acc[g][t]);
            }
          }
        }
      }

      // epilogue: lane holds channels nb .. nb+15 of pixel `row`, ordered [t][r]
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        if (nc0 + 64 * g >= N) continue;
        nt_epilogue<T>(ep, acc[g], row, rowvalid, nc0 + 64 * g + 16 * q, N, do_stats, sw, SWD, 64 * g + 16 * q, j);
      }
    }
  }

  if (do_stats) nt_flush_stats(ep, s_stat, SWD, nc0, N, rs, R, tid);
}


typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// ------------------------------------------------------------------------------------------------ gemm_nt, narrow (N <= 64, K <= 64)
// The stem and the first (non-expanding) block: 3.2e6 rows of 16..32 channels on either side.  The row-stationary kernel spends
// 4-8 weight-fragment loads and 128 cross-lane statistics operations on every 16-row tile that moves 1-2 KiB, and waits for each
// tile's activations before it starts the next.  Here the weight fragments and prologue coefficients of the lane's k positions
// stay in registers, the next tile's activations are in flight while the current one is computed, and the statistics are
// accumulated per lane and reduced once at the end (as in the column-stationary kernel).
// Round 6, NT > 0 (N a multiple of 4, bf16 output): the output tile of a lane is 4 channels x NT column tiles instead of 16 consecutive
// channels.  With the 16-channel form a lane group q owns channels 16 q .. 16 q + 15, so at N = 16 / 32 -- the only widths this kernel
// sees in the network -- a quarter / half of the lanes ran the epilogue (bias, residual, mask, rounding, statistics: most of the kernel's
// vector instructions, and the kernel issues 0.2 instructions per SIMD and cycle, close to what its waves can issue) while the others
// idled, and the wave computed four MFMA tiles of which one / two held outputs.  Now weight row 16 t + j feeds tile t, a lane holds
// channels 16 t + 4 q .. + 3 of pixel j (8-byte accesses, 512 contiguous bytes per wave and tile), only the NT tiles that exist are
// computed, and the per-channel vectors of the epilogue are read once.  Per channel the statistics add the same values in the same order
// (lane j over the row tiles, then the butterfly over j): bit-identical to NT = 0.
template <int MODE, int KST, int NT = 0>
__global__ __launch_bounds__(256) void k_gemm_nt_small(Operand A, const bf16_t* __restrict__ Wp, int ldw, Epilogue ep, long M, int N, int K) {
  using T = bf16_t;
  using MM = Mma<T>;
  constexpr int SWD = 64;
  constexpr int WT = NT > 0 ? NT : 4;   // MFMA tiles per k-step
  extern __shared__ float s_stat[];  // [4 waves][2][64] when statistics are taken
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int wrow = NT > 0 ? j : 16 * (j >> 2) + (j & 3);
  const bool do_stats = ep.stats != nullptr && ep.stat_mode != STAT_NONE;
  if (do_stats) {
    for (int i = tid; i < 8 * SWD; i += 256) s_stat[i] = 0.f;
    __syncthreads();
  }
  bf16x8 wf[KST][WT];
  float p1[KST][8], p2[KST][8], p3[KST][8];
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    const int k = ks * 32 + 8 * q;
#pragma unroll
    for (int t = 0; t < WT; ++t) wf[ks][t] = MM::raw(Wp + (long)(wrow + (NT > 0 ? 16 : 4) * t) * ldw + k);
#pragma unroll
    for (int e = 0; e < 8; ++e) p1[ks][e] = p2[ks][e] = p3[ks][e] = 0.f;
    if constexpr (MODE != PRO_NONE) {
      if (k < K) {   // per-channel vectors are readable up to K rounded up to 8
        VecIO<float, 8>::load(A.c1 + k, p1[ks]);
        VecIO<float, 8>::load(A.c2 + k, p2[ks]);
        if constexpr (MODE == PRO_BNBWD) VecIO<float, 8>::load(A.c3 + k, p3[ks]);
      }
    }
  }
  struct Raw { bf16x8 a, x; };
  Raw nx[KST];
  const long mtiles = (M + 15) / 16;
  const long stride = (long)gridDim.x * 4;
  auto fetch = [&](long mt) {
    const long row = mt * 16 + j;
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      const int k = ks * 32 + 8 * q;
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
      nx[ks].a = z;
      nx[ks].x = z;
      if (row < M && k < K) {
        nx[ks].a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(A.p1) + lay_off(row, k, A.ld1, A.ss1));
        if constexpr (MODE == PRO_BNBWD)
          nx[ks].x = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(A.p2) + lay_off(row, k, A.ld2, A.ss2));
      }
    }
  };
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
  // NT > 0: per-channel vectors of the lane's channels 16 t + 4 q + r (zero where the epilogue has none / past N)
  float e_bias[WT][4], e_zs[WT][4], e_zh[WT][4];
  const Act e_act = act_of(ep.mask);
  if constexpr (NT > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * t + 4 * q + r;
        e_bias[t][r] = (ep.bias && n < N) ? ep.bias[n] : 0.f;
        e_zs[t][r] = (ep.mask && n < N) ? ep.zscale[n] : 0.f;
        e_zh[t][r] = (ep.mask && n < N) ? ep.zshift[n] : 0.f;
      }
  }

  long mt = (long)blockIdx.x * 4 + wave;
  if (mt < mtiles) fetch(mt);
  for (; mt < mtiles; mt += stride) {
    Raw cur[KST];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) cur[ks] = nx[ks];
    if (mt + stride < mtiles) fetch(mt + stride);
    const long row = mt * 16 + j;
    const bool rowvalid = row < M;
    f32x4 acc[WT];
#pragma unroll
    for (int t = 0; t < WT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      const int k = ks * 32 + 8 * q;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = (float)cur[ks].a[e];
        if constexpr (MODE == PRO_NONE) v[e] = a;
        else if constexpr (MODE == PRO_BNRELU) v[e] = a * p1[ks][e] + p2[ks][e];
        else v[e] = p1[ks][e] * a + p2[ks][e] * (float)cur[ks].x[e] + p3[ks][e];
      }
      if constexpr (MODE == PRO_BNRELU) act_apply_v<8>(v, act_of(A.relu));
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (!rowvalid || k + e >= K) v[e] = 0.f;
      const bf16x8 af = MM::pack(v);
#pragma unroll
      for (int t = 0; t < WT; ++t) acc[t] = MM::mma(wf[ks][t], af, acc[t]);
    }
    if constexpr (NT > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c0 = 16 * t + 4 * q;
        float c[4], zv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = 0.f;
        if (rowvalid && c0 < N) {   // (N is a multiple of 4: the lane's four channels are valid together)
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = acc[t][r] + e_bias[t][r];
          if (ep.add) {
            const bf16x4 ad = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const T*>(ep.add) + row * ep.ldadd + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] += (float)ad[r];
          }
          if (ep.z) {
            const bf16x4 zb = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const T*>(ep.z) + lay_off(row, c0, ep.ldz, ep.zss));
#pragma unroll
            for (int r = 0; r < 4; ++r) zv[r] = (float)zb[r];
            if (ep.mask) {
              float a4[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) a4[r] = zv[r] * e_zs[t][r] + e_zh[t][r];
              act_bwd_v<4>(c, a4, e_act);
            }
          }
          bf16x4 ob;
#pragma unroll
          for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)c[r];
          u32x2 bits = __builtin_bit_cast(u32x2, ob);
          asm volatile("" : "+v"(bits));
          const f32x2 lo = bf16_pair_f32(bits[0]), hi = bf16_pair_f32(bits[1]);   // statistics see the stored values
          c[0] = lo[0]; c[1] = lo[1]; c[2] = hi[0]; c[3] = hi[1];
          *reinterpret_cast<u32x2*>(reinterpret_cast<T*>(ep.c) + lay_off(row, c0, ep.ldc, ep.css)) = bits;
        }
        if (do_stats) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[4 * t + r] += c[r];
            s2[4 * t + r] += (ep.stat_mode == STAT_SQ) ? c[r] * c[r] : c[r] * zv[r];
          }
        }
      }
    } else {
      float c[16], zv[16];
      nt_epilogue_core<T>(ep, acc, row, rowvalid, 16 * q, N, c, zv);
      if (do_stats) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s1[i] += c[i];
          s2[i] += (ep.stat_mode == STAT_SQ) ? c[i] * c[i] : c[i] * zv[i];
        }
      }
    }
  }
  if (do_stats) {
    float* sw = s_stat + wave * 2 * SWD;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (NT > 0 && i >= 4 * NT) continue;
      float a = s1[i], b = s2[i];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      const int ch = NT > 0 ? 16 * (i >> 2) + 4 * q + (i & 3) : 16 * q + i;
      if (j == 0 && ch < N) {
        sw[ch] = a;
        sw[SWD + ch] = b;
      }
    }
    nt_flush_stats(ep, s_stat, SWD, 0, N, blockIdx.x, gridDim.x, tid);
  }
}

// ------------------------------------------------------------------------------------------------ gemm_nt, column-stationary, streaming
// Round 3.  The first column-stationary kernel (k_gemm_nt_cs, removed in round 6) spent its time waiting, not moving bytes (3.3 TB/s on the 56x56 expand forward, where a bare write
// stream of the same shape and order reaches 5.3 TB/s, tools/probe/stpat.hip): s_waitcnt vmcnt counts loads AND stores in issue
// order, and every load or store that sits behind a branch (row / channel validity, the optional epilogue streams) makes the number
// of operations issued after a load unknown to the compiler, which then waits with vmcnt(0) -- i.e. for the PREVIOUS tile's stores
// to be acknowledged -- before it touches the operand of the current one.  Here every memory operation of the loop is unconditional:
//   * buffer loads / stores with a per-wave resource; an invalid lane (row >= M, k >= K, channel half outside the tensor) gets an
//     offset past num_records, so the hardware returns zeros / drops the store: no branch, and the zeros are exactly the padding the
//     MFMA and the statistics need;
//   * the operands of tile t+1 are issued before the arithmetic of tile t, the two stores of tile t after it: the wait for tile
//     t+1's operands is vmcnt(2), with the stores of tile t still in flight.
// The epilogue is fixed at compile time (EPK) instead of six run-time branches with their registers:
//   ST_FWD  : C = A * W^T, optional sum c / sum c^2 statistics               (the expand forward, mobilenet_base.py:316-320)
//   ST_MASK : C = mask_z(A * W^T), optional sum c / sum c*z statistics        (the input gradient of the projection through the
//             activation of the depthwise BatchNorm; A is the already differentiated BatchNorm output, see atomnas_bnbwd_apply)
//   ST_PBWD : ST_MASK plus the projection's weight gradient of the wave's 64 channels (atomnas_project_bwd without a prologue)
// A is plain bf16 without a prologue, K a multiple of 8; rows >= N and columns >= K of the packed weights Wp are zero.
constexpr int PB_RP = 16 + 4;   // rows of a tile + pad (elements): 40-byte rows, 8-byte aligned fragment reads
typedef __attribute__((ext_vector_type(4))) short s16x4;
enum { ST_FWD = 1, ST_MASK = 2, ST_PBWD = 3 };
// per k-steps: burst tiles of the narrow operand (0: per-tile ring), ring depth (of z, and of the operand when there is no
// burst), waves per SIMD the registers are allocated for
#ifndef ST_BT1
#define ST_BT1 16
#endif
#ifndef ST_BT2
#define ST_BT2 8
#endif
#ifndef ST_BT3
#define ST_BT3 4
#endif
#ifndef ST_SHARED
#define ST_SHARED 1
#endif
#ifndef ST_STORE_SOFFSET
#define ST_STORE_SOFFSET 0   // experiment builds: 1 = the tile offset of the output stores in an SGPR soffset (round 3/4 form)
#endif
#ifndef ST_BT6
#define ST_BT6 4   // six k-steps (K = 192): bursts only in the shared form (one tile is already a 6 KB request: per-tile ring otherwise)
#endif
#ifndef ST_PB_BT1
#define ST_PB_BT1 4
#endif
#ifndef ST_PB_BT2
#define ST_PB_BT2 2
#endif
#ifndef ST_WPE1
#define ST_WPE1 1
#endif
#ifndef ST_WPE2
#define ST_WPE2 1
#endif
#ifndef ST_WPE3
#define ST_WPE3 1
#endif
template <int KSTEPS, int EPK> struct StCfg {
  static constexpr int BT = EPK == ST_PBWD ? (KSTEPS == 1 ? ST_PB_BT1 : ST_PB_BT2)   // registers: + the weight-gradient accumulators
                                            : KSTEPS == 1 ? ST_BT1 : KSTEPS == 2 ? ST_BT2 : KSTEPS == 3 ? ST_BT3 : ST_BT6;
  static constexpr int PD = BT > 0 ? (BT >= 2 ? 2 : 1) : 2;
  // two waves per EU cap the allocation at 256 registers: the compiler then lets the MFMAs write VGPRs (no v_accvgpr_read per output:
  // -19 % instructions in the expand forward's loop, round 6); the six-k-step and the fused instances need more than 256
  static constexpr int WPE = EPK == ST_PBWD || KSTEPS == 6 ? 1 : 2;
  static constexpr bool SH = ST_SHARED && BT > 0 && BT % 4 == 0 && EPK != ST_PBWD;   // bursts shared by the workgroup's four waves
  static constexpr int BTP = KSTEPS <= 3 ? BT : 0;   // burst tiles of the private (per-wave) form
};
constexpr unsigned ST_OOB = 0x80000000u;   // >= num_records of every resource below

__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)ST_OOB, 0x00020000);
}
__device__ __forceinline__ bf16x8 st_as_bf16(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int KSTEPS, int EPK, int PD, int BT, int WPE, int UT, bool SH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_gemm_nt_st(Operand A, const bf16_t* __restrict__ Wp, int ldw, Epilogue ep, long M, int N, int K,
                                                    int nchunks, int tiles_per_item, float* __restrict__ ws) {
  using MM = Mma<bf16_t>;
  constexpr bool MASKED = EPK != ST_FWD;
  extern __shared__ u32x4 st_stage[];   // [4 waves][BT * KSTEPS][64] burst staging, then (ST_PBWD) [4 waves] transposed tiles
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, j = lane & 15;
  const int wrow = 16 * (j >> 2) + (j & 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long mtiles = (M + 15) / 16;
  const long nranges = (mtiles + tiles_per_item - 1) / tiles_per_item;
  int chunk;
  long range;
  bool wave_valid = true;
  if constexpr (SH) {
    // shared bursts: a workgroup is one row range x four consecutive chunks (grid = ranges x chunk groups); a wave past the last
    // chunk repeats it with its stores and statistics switched off -- it still loads its share of the bursts and meets the barriers
    const int ngroups = (nchunks + 3) / 4;
    range = blockIdx.x / ngroups;
    const int cr = (int)(blockIdx.x % ngroups) * 4 + wave;
    wave_valid = cr < nchunks;
    chunk = wave_valid ? cr : nchunks - 1;
  } else {
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= nranges * nchunks) return;
    chunk = (int)(item % nchunks);
    range = item / nchunks;
  }
  const int nc = chunk * 64;
  const int nb = nc + 16 * q;
  const bool do_stats = ep.stats != nullptr && wave_valid;
  constexpr int STAGE_U4 = (SH ? 2 : 4) * BT * KSTEPS * 64;   // burst staging in 16-byte units: two shared halves, or one slot per wave

  typename MM::frag wf[KSTEPS][4];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[ks][t] = MM::raw(Wp + (long)(nc + wrow + 4 * t) * ldw + ks * 32 + 8 * q);

  const long mt_beg = range * tiles_per_item;
  const long mt_end = mt_beg + tiles_per_item < mtiles ? mt_beg + tiles_per_item : mtiles;
  const int ntiles = (int)(mt_end - mt_beg);
  const long row0 = mt_beg * 16;
  const int rows_item = (int)((M - row0) < (long)ntiles * 16 ? (M - row0) : (long)ntiles * 16);

  // per-wave resources (base = first row of the range, first channel of the chunk) and per-lane byte offsets inside them
  const bf16_t* ap = reinterpret_cast<const bf16_t*>(A.p1) + (A.ss1 ? row0 * 16 : row0 * A.ld1);
  const __amdgpu_buffer_rsrc_t ra = st_rsrc(ap);
  const unsigned a_tile = A.ss1 ? 512u : 32u * (unsigned)A.ld1;
  unsigned a_lane[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int k = ks * 32 + 8 * q;
    const long e = A.ss1 ? (long)(k >> 4) * A.ss1 + j * 16 + (k & 15) : (long)j * A.ld1 + k;
    a_lane[ks] = k < K ? (unsigned)(2 * e) : ST_OOB;
  }
  bf16_t* cp = reinterpret_cast<bf16_t*>(ep.c) + (ep.css ? (long)(nc >> 4) * ep.css + row0 * 16 : row0 * ep.ldc + nc);
  const __amdgpu_buffer_rsrc_t rc = st_rsrc(cp);
  const unsigned c_tile = ep.css ? 512u : 32u * (unsigned)ep.ldc;
  const int nalloc = ep.css ? (N + 15) / 16 * 16 : N;
  unsigned c_lane[2], z_lane[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const long e = ep.css ? (long)q * ep.css + j * 16 + 8 * h : (long)j * ep.ldc + 16 * q + 8 * h;
    c_lane[h] = (nb + 8 * h < nalloc && wave_valid) ? (unsigned)(2 * e) : ST_OOB;
  }
  __amdgpu_buffer_rsrc_t rz = rc;
  unsigned z_tile = 0;
  // per-channel scale / shift of z: in a wave-private LDS slot (read back per tile), not in 32 registers
  float* s_zc = reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(st_stage + STAGE_U4) + 4 * (16 * (UT > 0 ? UT : 1) + 64) * PB_RP) + wave * 128;
  if constexpr (MASKED) {
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(ep.z) + (ep.zss ? (long)(nc >> 4) * ep.zss + row0 * 16 : row0 * ep.ldz + nc);
    rz = st_rsrc(zp);
    z_tile = ep.zss ? 512u : 32u * (unsigned)ep.ldz;
    const int zalloc = ep.zss ? (N + 15) / 16 * 16 : N;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long e = ep.zss ? (long)q * ep.zss + j * 16 + 8 * h : (long)j * ep.ldz + 16 * q + 8 * h;
      z_lane[h] = nb + 8 * h < zalloc ? (unsigned)(2 * e) : ST_OOB;
    }
    {   // per-channel vectors are readable up to N rounded up to 8; beyond that: zeros
      const int n = nc + lane;
      const bool ok = n < (N + 7) / 8 * 8;
      s_zc[lane] = ok ? ep.zscale[n] : 0.f;
      s_zc[64 + lane] = ok ? ep.zshift[n] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const Act am = act_of(ep.mask);

  f32x2 s1[8], s2[8];   // per-lane statistics of channels nb + 2 i, nb + 2 i + 1 (pairs: v_pk_add_f32 / v_pk_fma_f32)
#pragma unroll
  for (int i = 0; i < 8; ++i) s1[i] = s2[i] = f32x2{0.f, 0.f};
  // ST_PBWD: the weight gradient dWp[o][n] += sum_m A[m][o] * act(bn(z))[m][n] of the wave's 64 channels
  // (both operands transposed through a wave-private LDS region, 16x16x16 MFMAs over the tile's 16 rows)
  constexpr int UTA = UT > 0 ? UT : 1;
  f32x4 racc[UTA][4];
#pragma unroll
  for (int t = 0; t < UTA; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) racc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16_t* s_p = reinterpret_cast<bf16_t*>(st_stage + STAGE_U4) + wave * (16 * UTA + 64) * PB_RP;   // [16*UT][PB_RP] A^T
  bf16_t* s_a = s_p + 16 * UTA * PB_RP;                                                                         // [64][PB_RP] act(bn(z))^T

  // ---- epilogue of one tile: accumulators -> (mask) -> bf16 -> store, statistics of the stored values
  auto finish = [&](int t, const f32x4 (&acc)[4], const u32x4 (&zc)[2], const u32x4 (&araw)[KSTEPS]) {
    float c[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[4 * u + r] = acc[u][r];
    float zv[16], a16[16];
    if constexpr (MASKED) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x8 zb = st_as_bf16(zc[h]);
#pragma unroll
        for (int i = 0; i < 8; ++i) zv[8 * h + i] = (float)zb[i];
      }
      float zs[16], zh[16];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s_zc + 16 * q + 4 * v);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(s_zc + 64 + 16 * q + 4 * v);
#pragma unroll
        for (int r = 0; r < 4; ++r) { zs[4 * v + r] = a[r]; zh[4 * v + r] = b2[r]; }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) a16[i] = zv[i] * zs[i] + zh[i];
      if (__builtin_expect(am.swish, 0)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = nb + i < N ? c[i] * swish_grad(a16[i]) : 0.f;   // the padding of z may hold anything
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = act_pass(a16[i], am) ? c[i] : 0.f;
      }
    }
    if constexpr (EPK == ST_PBWD) {
      // the projection's forward operand act(bn(z)) and A, both transposed ([channel][row]); rows past M carry A = 0
      act_apply_v<16>(a16, am);
#pragma unroll
      for (int i = 0; i < 16; ++i) s_a[(16 * q + i) * PB_RP + j] = (bf16_t)a16[i];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8 af = st_as_bf16(araw[ks]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int o = ks * 32 + 8 * q + e;
          if (o < 16 * UTA) s_p[o * PB_RP + j] = af[e];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      s16x4 bfr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bfr[u] = *reinterpret_cast<const s16x4*>(s_a + (16 * u + j) * PB_RP + 4 * q);
#pragma unroll
      for (int tt = 0; tt < UTA; ++tt) {
        const s16x4 afr = *reinterpret_cast<const s16x4*>(s_p + (16 * tt + j) * PB_RP + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u) racc[tt][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(afr, bfr[u], racc[tt][u], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();   // the next tile overwrites s_p / s_a
    }
    const bool rv = j < rows_item - 16 * t;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 ob;
#pragma unroll
      for (int i = 0; i < 8; ++i) ob[i] = (bf16_t)c[8 * h + i];
      u32x4 obits = __builtin_bit_cast(u32x4, ob);
      asm volatile("" : "+v"(obits));   // the packed conversion happens once (common.h: bf16_pair_f32)
      // The tile offset rides in soffset (one scalar for all unrolled tiles instead of a vector induction variable per access and
      // tile).  HAZARD: with an SGPR soffset LLVM does not keep the next VALU instructions off the store's data registers (its rule
      // exempts that form), but on gfx950 data overwritten in the two instructions after the store IS what gets written (r03: 0.17 %
      // wrong elements in test_gemm_nt[20000-1440-80]).  The asm below uses `ob` after the statistics: its registers stay intact.
#if ST_STORE_SOFFSET
      __builtin_amdgcn_raw_buffer_store_b128(obits, rc, rv ? c_lane[h] : ST_OOB, (unsigned)t * c_tile, 0);
#else
      // round 5: the tile offset is added to the vector offset (soffset 0).  With the scalar offset the hazard above is NOT bounded by
      // two instructions: at M = 12544, N = 3456, K = 192 (the 7x7 expand, store queue backed up) the first data dword of a store was
      // replaced by a value written to that register ~28 instructions later (the next store's offset), 4 lanes at a time, ~4000 of 43 M
      // outputs per launch, different ones in every run (tools/gemmcheck.py, profiles/r05_st_store_hazard.txt).
      __builtin_amdgcn_raw_buffer_store_b128(obits, rc, rv ? c_lane[h] + (unsigned)t * c_tile : ST_OOB, 0, 0);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 o = bf16_pair_f32(obits[i]);   // statistics see the stored value
        s1[4 * h + i] += o;
        s2[4 * h + i] += EPK == ST_FWD ? o * o : o * f32x2{zv[8 * h + 2 * i], zv[8 * h + 2 * i + 1]};
        // the accumulation happens HERE: left to itself, instruction selection collects the statistics updates of all unrolled tiles at
        // the end of the loop body, with the 16 stored values of every tile live until then (about 20 registers per unrolled tile)
        asm volatile("" : "+v"(s1[4 * h + i]), "+v"(s2[4 * h + i]));
      }
      asm volatile("" ::"v"(obits));
    }
  };
  auto load_a = [&](int t, int ks) {
    const bool rv = j < rows_item - 16 * t;
    return __builtin_amdgcn_raw_buffer_load_b128(ra, rv ? a_lane[ks] : ST_OOB, (unsigned)t * a_tile, 0);
  };
  auto load_z = [&](int t, int h) {
    const bool rv = j < rows_item - 16 * t;
    return __builtin_amdgcn_raw_buffer_load_b128(rz, rv ? z_lane[h] : ST_OOB, (unsigned)t * z_tile, 0);
  };
  // Two ways to feed the narrow operand.  In both, tiles past the range (the loops run to a multiple of the unroll) and their
  // operands are all out of range: zeros in, nothing out.  A load still pending at the loop entry would put an s_waitcnt vmcnt(0)
  // into the loop header (the entry edge and the back edge share it), i.e. a wait for the previous tiles' stores in every iteration:
  // the first operands are waited for before the loop (the pins).  Pins also keep each tile's arithmetic behind the previous tile's
  // sched_barrier: instruction selection orders pure operations (the MFMAs) by data dependence only and had hoisted all unrolled
  // tiles' MFMAs to the top of the body (one set of accumulators and temporaries per tile).
  u32x4 zn[PD][2];
  if constexpr (BT > 0 && SH) {
    // (a') bursts SHARED by the workgroup's four waves (same row range, four chunks): the operand is read from L2 / HBM once per
    // workgroup instead of once per wave, in bursts of BT tiles of which every wave fetches a quarter (tiles wave, wave + 4, ...).
    // Two LDS halves: while the waves compute from one, the landed quarter-bursts are parked in the other and the next burst is
    // issued; one LDS-only barrier per BT tiles (s_barrier behind lgkmcnt(0): a __syncthreads() would also wait for every store).
    static_assert(BT % 4 == 0, "shared bursts: every wave fetches BT / 4 tiles");
    constexpr int NP = BT * KSTEPS;
    u32x4* sh = st_stage + lane;
    u32x4 ld[BT / 4][KSTEPS];
    auto burst = [&](int tb) {
#pragma unroll
      for (int iu = 0; iu < BT / 4; ++iu)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) ld[iu][ks] = load_a(tb + wave + 4 * iu, ks);
    };
    auto park = [&](int half) {
#pragma unroll
      for (int iu = 0; iu < BT / 4; ++iu)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          asm volatile("" : "+v"(ld[iu][ks]));
          sh[(half * NP + (wave + 4 * iu) * KSTEPS + ks) * 64] = ld[iu][ks];
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    burst(0);
    if constexpr (MASKED) {
#pragma unroll
      for (int u = 0; u < PD; ++u) { zn[u][0] = load_z(u, 0); zn[u][1] = load_z(u, 1); }
#pragma unroll
      for (int u = 0; u < PD; ++u) { asm volatile("" : "+v"(zn[u][0])); asm volatile("" : "+v"(zn[u][1])); }
    }
    park(0);
    burst(BT);
    lds_barrier();
    int half = 0;
    for (int t0 = 0; t0 < ntiles; t0 += BT) {
#pragma unroll
      for (int u = 0; u < BT; ++u) {
        int t = t0 + u;
        asm volatile("" : "+s"(t));
        u32x4 ac[KSTEPS], zc[2];
        unsigned eo = (unsigned)((half * NP + u * KSTEPS) * 64);
        asm volatile("" : "+v"(eo));
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          ac[ks] = sh[eo + ks * 64];
          asm volatile("" : "+v"(ac[ks]));
        }
        if constexpr (MASKED) {
          asm volatile("" : "+v"(zn[u % PD][0]));
          zc[0] = zn[u % PD][0]; zc[1] = zn[u % PD][1];
        }
        f32x4 acc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const bf16x8 af = st_as_bf16(ac[ks]);
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = MM::mma(wf[ks][v], af, acc[v]);
        }
        finish(t, acc, zc, ac);
        if constexpr (MASKED) { zn[u % PD][0] = load_z(t + PD, 0); zn[u % PD][1] = load_z(t + PD, 1); }
        __builtin_amdgcn_sched_barrier(0);
      }
      int tb = t0 + 2 * BT;
      asm volatile("" : "+s"(tb));
      park(half ^ 1);      // the burst issued one period ago; that half was last read before the previous barrier
      burst(tb);
      lds_barrier();
      half ^= 1;
    }
  } else if constexpr (BT > 0) {
    // (a) bursts through LDS.  HBM reads trickling into a saturated write stream are what this kernel pays for (r03 knock-outs on
    // the 56x56 expand, 694 MB out / 38 MB in: 215 us; narrow operand from an L2-resident window 157 us; one request burst per
    // 4 / 8 tiles 198 / 171 us): the operand is fetched BT tiles (several KB, contiguous) at a time.  The burst lands in registers,
    // is parked in a lane-private LDS slot (each lane reads back exactly what it wrote: no barrier), and the same registers take the
    // next burst, which has the whole BT-tile period to arrive.
    u32x4* my = st_stage + (long)wave * (BT * KSTEPS * 64) + lane;
    u32x4 bn[BT][KSTEPS];
#pragma unroll
    for (int u = 0; u < BT; ++u)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) bn[u][ks] = load_a(u, ks);
    if constexpr (MASKED) {
#pragma unroll
      for (int u = 0; u < PD; ++u) { zn[u][0] = load_z(u, 0); zn[u][1] = load_z(u, 1); }
#pragma unroll
      for (int u = 0; u < PD; ++u) { asm volatile("" : "+v"(zn[u][0])); asm volatile("" : "+v"(zn[u][1])); }
    }
#pragma unroll
    for (int u = 0; u < BT; ++u)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(bn[u][ks]));
    for (int t0 = 0; t0 < ntiles; t0 += BT) {
      int tb = t0 + BT;
      asm volatile("" : "+s"(tb));
#pragma unroll
      for (int u = 0; u < BT; ++u)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          asm volatile("" : "+v"(bn[u][ks]));
          my[(u * KSTEPS + ks) * 64] = bn[u][ks];
        }
#pragma unroll
      for (int u = 0; u < BT; ++u)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) bn[u][ks] = load_a(tb + u, ks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < BT; ++u) {
        int t = t0 + u;
        asm volatile("" : "+s"(t));   // opaque per tile: validity masks and offsets of all unrolled tiles are otherwise computed up front
        u32x4 ac[KSTEPS], zc[2];
        unsigned eo = (unsigned)(u * KSTEPS * 64);
        asm volatile("" : "+v"(eo));   // the LDS reads of all unrolled tiles would otherwise be hoisted to the top of the body
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          ac[ks] = my[eo + ks * 64];
          asm volatile("" : "+v"(ac[ks]));
        }
        if constexpr (MASKED) {
          asm volatile("" : "+v"(zn[u % PD][0]));
          zc[0] = zn[u % PD][0]; zc[1] = zn[u % PD][1];
        }
        f32x4 acc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const bf16x8 af = st_as_bf16(ac[ks]);
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = MM::mma(wf[ks][v], af, acc[v]);
        }
        finish(t, acc, zc, ac);
        if constexpr (MASKED) { zn[u % PD][0] = load_z(t + PD, 0); zn[u % PD][1] = load_z(t + PD, 1); }
        __builtin_amdgcn_sched_barrier(0);   // one tile at a time
      }
    }
  } else {
    // (b) a ring of PD register sets, one tile refilled per tile (the wide-K instances: one tile is already a 6 KB request).  The
    // wait for tile t+1's operands leaves the stores of the PD-1 tiles before it in flight.  The pin is on the ring registers
    // themselves, and the refill of a slot is issued after the MFMAs that read it: with a pinned COPY the refill went to other
    // registers and the ring was rotated with moves behind a wait for every load.
    u32x4 an[PD][KSTEPS];
#pragma unroll
    for (int u = 0; u < PD; ++u) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) an[u][ks] = load_a(u, ks);
      if constexpr (MASKED) { zn[u][0] = load_z(u, 0); zn[u][1] = load_z(u, 1); }
    }
#pragma unroll
    for (int u = 0; u < PD; ++u) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(an[u][ks]));
      if constexpr (MASKED) { asm volatile("" : "+v"(zn[u][0])); asm volatile("" : "+v"(zn[u][1])); }
    }
    for (int t0 = 0; t0 < ntiles; t0 += PD) {
#pragma unroll
      for (int slot = 0; slot < PD; ++slot) {
        int t = t0 + slot;
        asm volatile("" : "+s"(t));
        u32x4 zc[2];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(an[slot][ks]));
        if constexpr (MASKED) { zc[0] = zn[slot][0]; zc[1] = zn[slot][1]; }
        f32x4 acc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const bf16x8 af = st_as_bf16(an[slot][ks]);
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = MM::mma(wf[ks][v], af, acc[v]);
        }
        u32x4 akeep[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          akeep[ks] = an[slot][ks];
          an[slot][ks] = load_a(t + PD, ks);
        }
        if constexpr (MASKED) { zn[slot][0] = load_z(t + PD, 0); zn[slot][1] = load_z(t + PD, 1); }
        finish(t, acc, zc, akeep);
        __builtin_amdgcn_sched_barrier(0);   // one tile at a time
      }
    }
  }

  if (do_stats) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float a = s1[i >> 1][i & 1], b = s2[i >> 1][i & 1];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      if (j == 0 && nb + i < N) {   // this wave is the only writer of (row `range`, channel nb + i): plain stores
        ep.stats[(long)range * 2 * N + nb + i] = a;
        ep.stats[(long)range * 2 * N + N + nb + i] = b;
        stat_zero_tail(ep.stats, 2L * N, (int)(range + nranges), (int)nranges, ep.stat_rows, nb + i);
        stat_zero_tail(ep.stats, 2L * N, (int)(range + nranges), (int)nranges, ep.stat_rows, (long)N + nb + i);
      }
    }
  }
  if (EPK == ST_PBWD && wave_valid) {
    // partial of the weight gradient: (o, n) at ws[(range * K + o) * N + n], summed in range order by reduce_parts
#pragma unroll
    for (int t = 0; t < UTA; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = nc + 16 * u + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * t + 4 * q + r;
          if (o < K && n < N) ws[((long)range * K + o) * N + n] = racc[t][u][r];
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ gemm_nt, weights shared in LDS
// For the contraction-heavy shapes (K > 192: the linear projection forward, the expand input-gradient): the 6x-wide hidden
// tensor is the INPUT, streamed once from HBM straight into MFMA B fragments (prologue applied in registers).  In the
// row-stationary kernel every wave re-loads the weight fragments and the prologue coefficients of every k-step from
// L1/L2 -- 8..12 vector-memory instructions per 1 KiB of activations, which bounds it at the texture-address rate, not at
// HBM.  Here a workgroup owns 64*RT rows, and the weights (64*NCG channels x 64 k) and coefficients of a k-chunk are
// staged ONCE per workgroup into LDS (double-buffered, one barrier per chunk) and read back as ds_read_b128 fragments:
// the only vector-memory traffic left in the k-loop is the activation stream itself, prefetched one chunk ahead.
#ifndef WS_TIMING
#define WS_TIMING 0   // s_memtime phase accounting of k_gemm_nt_ws (tools/wstiming.py, experiment builds only)
#endif
#if WS_TIMING
__device__ unsigned long long g_ws_timing[8];
#define WS_MARK(i) { const unsigned long long t_ = __builtin_readcyclecounter(); wacc[i] += t_ - wlast; wlast = t_; }
#else
#define WS_MARK(i)
#endif
constexpr int WS_KC = 64;            // k per chunk (two MFMA k-steps)
constexpr int WS_WP = WS_KC + 8;     // LDS row pitch (elements): 16 consecutive rows -> 16 distinct 16-byte bank groups

template <int MODE> struct WsRaw { bf16x8 a; };
template <> struct WsRaw<PRO_BNBWD> { bf16x8 a, x; };

template <int MODE, int NCG, int RT>
__global__ __launch_bounds__(256) void k_gemm_nt_ws(Operand A, const bf16_t* __restrict__ Wp, int ldw, int wrows, Epilogue ep, long M,
                                                    int N, int K) {
  using T = bf16_t;
  using MM = Mma<T>;
  constexpr int WROWS = 64 * NCG;
  constexpr int WBUF = WROWS * WS_WP;
  constexpr int NVEC = (MODE == PRO_BNRELU) ? 2 : (MODE == PRO_BNBWD ? 3 : 0);
  constexpr int WPASS = WROWS * 8 / 256;   // 16-byte weight pieces per thread per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ws[];
  T* s_w = reinterpret_cast<T*>(smem_ws);                     // [2][WROWS][WS_WP], rows permuted (see below)
  float* s_c = reinterpret_cast<float*>(s_w + 2 * WBUF);      // [2][3][WS_KC]
  float* s_stat = s_c + 2 * 3 * WS_KC;                        // [4 waves][2][WROWS]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  const bool do_stats = ep.stats != nullptr && ep.stat_mode != STAT_NONE;
#if WS_TIMING
  unsigned long long wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long wlast = __builtin_readcyclecounter();
#endif
  if (do_stats) {
    for (int i = tid; i < 8 * WROWS; i += 256) s_stat[i] = 0.f;
  }
  float* sw = s_stat + wave * 2 * WROWS;
  const int nchunk = (K + WS_KC - 1) / WS_KC;
  const int K8 = (K + 7) & ~7;
  const long rblocks = (M + 64 * RT - 1) / (64 * RT);
  const int ngroups = (N + WROWS - 1) / WROWS;

  // staging role: piece p of this thread is weight row (tid + 256 p) / 8 of the group, 16-byte segment tid % 8.  MFMA tile t of
  // a 64-channel chunk uses channels 16*(jj>>2) + 4*t + (jj&3) for its 16 rows jj: they are stored as LDS rows t*16 + jj, so
  // that the 16 rows one fragment read touches are consecutive (conflict-free at pitch 72).
  const int sseg = tid & 7;
  int srow_g[WPASS], srow_l[WPASS];
#pragma unroll
  for (int p = 0; p < WPASS; ++p) {
    const int r = (tid + 256 * p) >> 3;
    const int g = r >> 6, n = r & 63;
    srow_g[p] = r;
    srow_l[p] = g * 64 + ((n >> 2) & 3) * 16 + (((n >> 4) << 2) | (n & 3));
  }
  // (a second chunk of weights in flight -- staging loads issued two iterations before their LDS store -- measured no gain on the
  // projection and -3..-9 % on the two-stream input gradient, r03: the chunk iteration is not waiting for these loads)
  bf16x8 wA[WPASS];
  f32x4 cA = f32x4{0.f, 0.f, 0.f, 0.f};
  // Round 3: every load of the k-loop is UNCONDITIONAL (addresses clamped into the tensor, the value replaced by zero where the
  // original condition failed).  With loads behind branches the compiler cannot count the operations issued after a load and waits
  // with s_waitcnt vmcnt(0): the activation prefetch below could then never be more than one chunk deep.
  auto stage_load = [&](int nc0, int c, bf16x8 (&wreg)[WPASS], f32x4& creg) {
    const int k = c * WS_KC + sseg * 8;
    const int kc = k < ldw ? k : ldw - 8;
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
#pragma unroll
    for (int p = 0; p < WPASS; ++p) {
      const int wr = nc0 + srow_g[p] < wrows ? nc0 + srow_g[p] : wrows - 1;
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(Wp + (long)wr * ldw + kc);
      wreg[p] = (k < ldw && nc0 + srow_g[p] < wrows) ? w : z;
    }
    if constexpr (NVEC > 0) {
      const int v = (tid >> 4) < NVEC ? (tid >> 4) : 0, kk = c * WS_KC + (tid & 15) * 4;
      const float* src = (v == 0) ? A.c1 : (v == 1 ? A.c2 : A.c3);
      const f32x4 cv = *reinterpret_cast<const f32x4*>(src + (kk < K8 ? kk : K8 - 4));
      creg = kk < K8 ? cv : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stage_store = [&](int buf, const bf16x8 (&wreg)[WPASS], const f32x4& creg) {
#pragma unroll
    for (int p = 0; p < WPASS; ++p) *reinterpret_cast<bf16x8*>(s_w + buf * WBUF + srow_l[p] * WS_WP + sseg * 8) = wreg[p];
    if constexpr (NVEC > 0) {
      if (tid < NVEC * 16) *reinterpret_cast<f32x4*>(s_c + buf * 3 * WS_KC + (tid >> 4) * WS_KC + (tid & 15) * 4) = creg;
    }
  };

  // fixed channel group per workgroup, row blocks rs, rs+R, ... (see k_gemm_nt)
  const int grp = blockIdx.x % ngroups, rs = blockIdx.x / ngroups, R = gridDim.x / ngroups;
  const int nc0 = grp * WROWS;
  // Raw activations of one chunk: [subtile][k-step], 8 consecutive k of one pixel per lane.  The stream is prefetched TWO chunks
  // ahead, across row-block boundaries (the sequence of (row block, chunk) items of this workgroup is one pipeline): with one
  // chunk in flight per wave the kernel was latency-bound -- 2.3 TB/s on the 56x56 projection, 0.9 TB/s on the 7x7 one (54 chunks
  // of one round trip each), while the same kernel with two streams (the BatchNorm-backward prologue) reached 4.7 TB/s.
  // Rows past M and k past K read clamped addresses: their products meet zero weights / are dropped by the epilogue.
  constexpr bool LANE_STATS = NCG == 1;
  float ls1[16], ls2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ls1[i] = ls2[i] = 0.f;
  WsRaw<MODE> a1[RT][2], a2[RT][2];
  auto load_a = [&](long rb, int c, WsRaw<MODE> (&dst)[RT][2]) {
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      long r = rb * (64 * RT) + (wave * RT + s) * 16 + j;
      r = r < M ? r : M - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        int k = c * WS_KC + ks * 32 + 8 * q;
        k = k < K ? k : K8 - 8;
        dst[s][ks].a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(A.p1) + lay_off(r, k, A.ld1, A.ss1));
        if constexpr (MODE == PRO_BNBWD)
          dst[s][ks].x = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(A.p2) + lay_off(r, k, A.ld2, A.ss2));
      }
    }
  };
  stage_load(nc0, 0, wA, cA);
  load_a(rs, 0, a1);
  if (nchunk > 1) load_a(rs, 1, a2); else load_a(rs + R, 0, a2);
  stage_store(0, wA, cA);
  __syncthreads();
  int buf = 0;
  for (long rb = rs; rb < rblocks; rb += R) {
    long row[RT];
    bool rowvalid[RT];
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      row[s] = rb * (64 * RT) + (wave * RT + s) * 16 + j;
      rowvalid[s] = row[s] < M;
    }
    f32x4 acc[RT][NCG][4];
#pragma unroll
    for (int s = 0; s < RT; ++s)
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][g][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nchunk; ++c) {
      WS_MARK(0)   // loop top / epilogue of the previous block
      WsRaw<MODE> acur[RT][2];
#pragma unroll
      for (int s = 0; s < RT; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { acur[s][ks] = a1[s][ks]; a1[s][ks] = a2[s][ks]; }
#if WS_TIMING
#pragma unroll
      for (int s = 0; s < RT; ++s) { asm volatile("" : "+v"(acur[s][0].a)); asm volatile("" : "+v"(acur[s][1].a)); }
      WS_MARK(1)   // wait for this chunk's activations
#endif
      {   // weights of the next item, activations of the one after it
        stage_load(nc0, c + 1 < nchunk ? c + 1 : 0, wA, cA);
        int c2 = c + 2;
        long rb2 = rb;
        if (c2 >= nchunk) { c2 -= nchunk; rb2 += R; }
        if (c2 >= nchunk) { c2 -= nchunk; rb2 += R; }   // nchunk == 1
        load_a(rb2, c2, a2);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kl = ks * 32 + 8 * q;
        float c1v[8], c2v[8], c3v[8];
        if constexpr (NVEC > 0) {
          const float* cb = s_c + buf * 3 * WS_KC;
          VecIO<float, 8>::load(cb + kl, c1v);
          VecIO<float, 8>::load(cb + WS_KC + kl, c2v);
          if constexpr (NVEC > 2) VecIO<float, 8>::load(cb + 2 * WS_KC + kl, c3v);
        }
        bf16x8 af[RT];
#pragma unroll
        for (int s = 0; s < RT; ++s) {
          if constexpr (MODE == PRO_NONE) {
            af[s] = acur[s][ks].a;
          } else {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float a = (float)acur[s][ks].a[e];
              if constexpr (MODE == PRO_BNRELU) v[e] = a * c1v[e] + c2v[e];
              else v[e] = c1v[e] * a + c2v[e] * (float)acur[s][ks].x[e] + c3v[e];
            }
            if constexpr (MODE == PRO_BNRELU) act_apply_v<8>(v, act_of(A.relu));
            af[s] = MM::pack(v);
          }
        }
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(s_w + buf * WBUF + (g * 64 + t * 16 + j) * WS_WP + kl);
#pragma unroll
            for (int s = 0; s < RT; ++s) acc[s][g][t] = MM::mma(wf, af[s], acc[s][g][t]);
          }
      }
#if WS_TIMING
#pragma unroll
      for (int s = 0; s < RT; ++s)
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc[s][g][t]));
      WS_MARK(2)   // issue of the next loads, prologue, fragment reads, MFMAs
#endif
      stage_store(buf ^ 1, wA, cA);
#if WS_TIMING
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WS_MARK(3)   // wait for the staged weights, LDS store
#endif
      __syncthreads();
      WS_MARK(4)   // barrier
      buf ^= 1;
    }

#pragma unroll
    for (int s = 0; s < RT; ++s)
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        if (nc0 + 64 * g >= N) continue;
        if constexpr (LANE_STATS) {
          // one chunk group (N <= 64): the statistics stay per lane across the row blocks and are reduced over the 16 rows of a
          // tile once, after the loop -- the per-tile reduction (128 cross-lane operations) was 28 % of the wave's cycles on the
          // 56x56 projection (tools/wstiming.py, r03)
          float c[16], zv[16];
          nt_epilogue_core<T>(ep, acc[s][g], row[s], rowvalid[s], nc0 + 16 * q, N, c, zv);
          if (do_stats) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              ls1[i] += c[i];
              ls2[i] += (ep.stat_mode == STAT_SQ) ? c[i] * c[i] : c[i] * zv[i];
            }
          }
        } else {
          nt_epilogue<T>(ep, acc[s][g], row[s], rowvalid[s], nc0 + 64 * g + 16 * q, N, do_stats, sw, WROWS, 64 * g + 16 * q, j);
        }
      }
    WS_MARK(5)   // epilogue
  }
  if constexpr (LANE_STATS) {
    if (do_stats) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float a = ls1[i], b = ls2[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        if (j == 0 && nc0 + 16 * q + i < N) {
          sw[16 * q + i] += a;
          sw[WROWS + 16 * q + i] += b;
        }
      }
    }
  }

  WS_MARK(5)   // epilogue of the last block
  if (do_stats) nt_flush_stats(ep, s_stat, WROWS, nc0, N, rs, R, tid);
#if WS_TIMING
  WS_MARK(6)
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) atomicAdd(&g_ws_timing[i], wacc[i]);
  }
#endif
}

// ------------------------------------------------------------------------------------------------ fused expand backward
// Backward of the expand 1x1 convolution (ConvBNReLU(inp, hid, 1), models/mobilenet_base.py:316-320) for the early stages, where
// inp is 16..48 and the hidden tensors are the whole cost: the input gradient dX = dE * We and the weight gradient dWe = dE^T x
// both need dE = c1*h + c2*Eraw + c3 (the BatchNorm backward of the two hidden streams h and Eraw).  As two GEMMs (gemm_nt with
// the BNBWD prologue + gemm_tn) the two streams are read twice; here they are read ONCE:
//   * k_gemm_nt_ws's structure for dX: a workgroup owns 64-row blocks, streams h / Eraw from HBM straight into MFMA B fragments
//     (prologue in registers), the weights of a 64-channel chunk come from LDS;
//   * the same fragments are written transposed into LDS ([channel][row], double-buffered) and, one chunk later, multiplied with
//     the block's x^T (staged once per row block): R[inp][64 channels] += x^T dE, accumulated in registers over the workgroup's
//     rows for ALL chunks (NCH x UT accumulator tiles per wave: the chunk loop is unrolled so that their indices are static);
//   * per-workgroup partials of R in the caller's workspace, summed in workgroup order by reduce_parts (bit-reproducible).
#ifndef XB_MINB2_LIMIT
#define XB_MINB2_LIMIT 64   // accumulator registers (4 * UT * NCH) up to which the kernel is compiled for two workgroups per CU
#endif
constexpr int XB_DP = 64 + 8;   // transposed LDS pitch (elements): rows of the block + pad; 144 bytes, 16-byte aligned rows

#ifndef EB_MINWG
#define EB_MINWG 2   // workgroups per CU the registers are allocated for (experiment switch)
#endif
// dE = c1*h only (round 4; the two-stream form that read the raw expand output E as well was removed in round 6): the c2*E + c3 part
// of the BatchNorm backward reaches dX / dWe through inp x inp sized corrections (atomnas_xb_coeffs), so E is not read.
template <int UT, int NCH>
__global__ __launch_bounds__(256, EB_MINWG) void k_expand_bwd(Operand A, const bf16_t* __restrict__ Wp, int ldw, int wrows, const bf16_t* __restrict__ x,
                                                    int ldx, Epilogue ep, float* __restrict__ ws, long M, int N, int K,
                                                    const bf16_t* __restrict__ mpk, int ldm) {
  using T = bf16_t;
  using MM = Mma<T>;
  constexpr int WROWS = 64;
  constexpr int WBUF = WROWS * WS_WP;
  constexpr int WPASS = WROWS * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_xb[];
  T* s_w = reinterpret_cast<T*>(smem_xb);                      // [2][64][WS_WP] weights of a chunk, rows permuted (k_gemm_nt_ws)
  float* s_c = reinterpret_cast<float*>(s_w + 2 * WBUF);       // [2][3][WS_KC] BatchNorm-backward coefficients of a chunk
  T* s_d = reinterpret_cast<T*>(s_c + 2 * 3 * WS_KC);          // [2][64 channels][XB_DP] dE of a chunk, transposed
  T* s_x = s_d + 2 * 64 * XB_DP;                               // [16*UT][XB_DP] x of the row block, transposed

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int nchunk = (K + WS_KC - 1) / WS_KC;
  const int K8 = (K + 7) & ~7;
  const long rblocks = (M + 63) / 64;

  const int sseg = tid & 7;
  int srow_g[WPASS], srow_l[WPASS];
#pragma unroll
  for (int p = 0; p < WPASS; ++p) {
    const int r = (tid + 256 * p) >> 3;
    const int n = r & 63;
    srow_g[p] = r;
    srow_l[p] = ((n >> 2) & 3) * 16 + (((n >> 4) << 2) | (n & 3));
  }
  bf16x8 wreg[WPASS];
  f32x4 creg = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage_load = [&](int c) {
    const int k = c * WS_KC + sseg * 8;
#pragma unroll
    for (int p = 0; p < WPASS; ++p) {
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
      wreg[p] = z;
      if (k < ldw && srow_g[p] < wrows) wreg[p] = *reinterpret_cast<const bf16x8*>(Wp + (long)srow_g[p] * ldw + k);
    }
    if (tid < 3 * 16) {
      const int v = tid >> 4, kk = c * WS_KC + (tid & 15) * 4;
      const float* src = (v == 0) ? A.c1 : (v == 1 ? A.c2 : A.c3);
      creg = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kk < K8) creg = *reinterpret_cast<const f32x4*>(src + kk);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int p = 0; p < WPASS; ++p) *reinterpret_cast<bf16x8*>(s_w + buf * WBUF + srow_l[p] * WS_WP + sseg * 8) = wreg[p];
    if (tid < 3 * 16) *reinterpret_cast<f32x4*>(s_c + buf * 3 * WS_KC + (tid >> 4) * WS_KC + (tid & 15) * 4) = creg;
  };

  f32x4 racc[NCH][UT];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < UT; ++t) racc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int rs = blockIdx.x, R = gridDim.x;
  for (long rb = rs; rb < rblocks; rb += R) {
    const long row = rb * 64 + wave * 16 + j;
    const bool rowvalid = row < M;
    // x of the block, transposed: piece = (row r, 8 channels cg); channels >= N and rows >= M are zero
    for (int idx = tid; idx < 64 * 2 * UT; idx += 256) {
      const int r = idx & 63, cg = idx >> 6;
      const long xr = rb * 64 + r;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
      if (xr < M && cg * 8 < ((N + 7) & ~7)) v = *reinterpret_cast<const bf16x8*>(x + xr * ldx + cg * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s_x[(cg * 8 + e) * XB_DP + r] = (cg * 8 + e < N) ? v[e] : (bf16_t)0.f;
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    WsRaw<PRO_BNBWD> anx[2], acur[2];
    auto load_a = [&](int c) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k = c * WS_KC + ks * 32 + 8 * q;
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
        anx[ks].a = z;
        anx[ks].x = z;
        if (rowvalid && k < K) {
          anx[ks].a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const T*>(A.p1) + lay_off(row, k, A.ld1, A.ss1));
        }
      }
    };
    // weight-gradient product of chunk cc (its transposed dE tile is complete: a barrier has passed since it was written)
    auto wgrad = [&](int cc, f32x4 (&ra)[UT]) {
      const T* d = s_d + (cc & 1) * 64 * XB_DP;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const bf16x8 db = *reinterpret_cast<const bf16x8*>(d + (16 * wave + j) * XB_DP + 32 * k2 + 8 * q);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          const bf16x8 xa = *reinterpret_cast<const bf16x8*>(s_x + (16 * t + j) * XB_DP + 32 * k2 + 8 * q);
          ra[t] = MM::mma(xa, db, ra[t]);
        }
      }
    };

    stage_load(0);
    load_a(0);
    stage_store(0);
    __syncthreads();
#pragma unroll
    for (int c = 0; c <= NCH; ++c) {
      if (c > 0 && c <= nchunk) wgrad(c - 1, racc[c > 0 ? c - 1 : 0]);
      if (c < NCH && c < nchunk) {
        const int buf = c & 1;
        acur[0] = anx[0];
        acur[1] = anx[1];
        if (c + 1 < nchunk) {
          stage_load(c + 1);
          load_a(c + 1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int kl = ks * 32 + 8 * q;
          float c1v[8], c2v[8], c3v[8];
          const float* cb = s_c + buf * 3 * WS_KC;
          VecIO<float, 8>::load(cb + kl, c1v);
          VecIO<float, 8>::load(cb + WS_KC + kl, c2v);
          VecIO<float, 8>::load(cb + 2 * WS_KC + kl, c3v);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = rowvalid ? c1v[e] * (float)acur[ks].a[e] : 0.f;
          const bf16x8 af = MM::pack(v);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(s_w + buf * WBUF + (t * 16 + j) * WS_WP + kl);
            acc[t] = MM::mma(wf, af, acc[t]);
          }
          T* d = s_d + buf * 64 * XB_DP + kl * XB_DP + wave * 16 + j;
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e * XB_DP] = af[e];
        }
        if (c + 1 < nchunk) stage_store(buf ^ 1);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);   // the chunk loop is unrolled for static accumulator indices only: no motion across chunks
      }
    }
    {
      // + x M (mpk: M = We^T diag(c2) We packed [N rounded up to 64][ldm], zero padded): the c2 term of the BatchNorm backward, a
      // product of the narrow input with an inp x inp matrix; its bias v rides in the epilogue
      if (mpk) {
        for (int kx = 0; kx < N; kx += 32) {
          const int k = kx + 8 * q;
          bf16x8 xb;
#pragma unroll
          for (int e = 0; e < 8; ++e) xb[e] = (bf16_t)0.f;
          if (rowvalid && k < N) xb = *reinterpret_cast<const bf16x8*>(x + row * ldx + k);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int n = 16 * (j >> 2) + 4 * t + (j & 3);   // output channel of accumulator row j of tile t (row permutation of s_w)
            const bf16x8 mf = *reinterpret_cast<const bf16x8*>(mpk + (long)n * ldm + k);
            acc[t] = MM::mma(mf, xb, acc[t]);
          }
        }
      }
    }
    nt_epilogue<T>(ep, acc, row, rowvalid, 16 * q, N, false, nullptr, 0, 0, j);
    __syncthreads();   // s_x and the last dE tile are free for the next row block
  }

  // this workgroup's partial of R = x^T dE: element (x channel uc, hidden channel vc) at ws[(rs * N + uc) * K + vc]
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int vc = c * 64 + 16 * wave + j;
    if (c < nchunk && vc < K) {
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int uc = 16 * t + 4 * q + r;
          if (uc < N) ws[((long)rs * N + uc) * K + vc] = racc[c][t][r];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
// The streaming instances take the cases they are written for (see k_gemm_nt_st); everything else goes on to the LDS-weights / generic kernels.
static int nt_st_kind(int mode, const Operand& A, const Epilogue& ep, long M, int N, int K) {
  static const int on = getenv("ATOMNAS_NT_ST") ? atoi(getenv("ATOMNAS_NT_ST")) : 1;
  if (!on || mode != PRO_NONE || (K & 7) || ep.bias || ep.add || ep.out_f32) return 0;
  const bool stats = ep.stats != nullptr && ep.stat_mode != STAT_NONE;
  if (ep.stats != nullptr && ep.stat_mode == STAT_NONE) return 0;
  // a row range is at most 4096 tiles (32-bit tile offsets) and owns one partial row of the statistics
  if (stats && ((M + 15) / 16 + ep.stat_rows - 1) / ep.stat_rows > 4096) return 0;
  // 32-bit byte offsets below 2^31 inside the per-wave resources: the whole A tensor, four slabs (one chunk) of C and z
  const long a_bytes = A.ss1 ? ((long)((K + 15) / 16 - 1) * A.ss1 + M * 16) * 2 : M * (long)A.ld1 * 2;
  if (a_bytes >= (1L << 31)) return 0;
  if (ep.css ? (3 * ep.css + M * 16) * 2 >= (1L << 31) : ((N & 7) || (long)ep.ldc * 2 * 16 * 4096 >= (1L << 31))) return 0;
  if (!ep.z) return (!stats || ep.stat_mode == STAT_SQ) ? ST_FWD : 0;
  if (!ep.mask || (stats && ep.stat_mode != STAT_Z)) return 0;
  if (ep.zss ? (3 * ep.zss + M * 16) * 2 >= (1L << 31) : ((N & 7) || (long)ep.ldz * 2 * 16 * 4096 >= (1L << 31))) return 0;
  return ST_MASK;
}

template <int KSTEPS>
static void launch_nt_st(int kind, const Operand& A, const void* Wp, int ldw, const Epilogue& ep, long M, int N, int K, hipStream_t st) {
  const int nchunks = (N + 63) / 64;
  const long mtiles = (M + 15) / 16;
  const bf16_t* W = (const bf16_t*)Wp;
  const long min_tpi = ep.stats ? (mtiles + ep.stat_rows - 1) / ep.stat_rows : 1;   // every row range owns one partial row
#define ST_LAUNCH(EPKV, SHV, BTV)                                                                                         \
  {                                                                                                                     \
    using Cfg = StCfg<KSTEPS, EPKV>;                                                                                    \
    auto kern = k_gemm_nt_st<KSTEPS, EPKV, Cfg::PD, BTV, Cfg::WPE, 0, SHV>;                                            \
    const size_t lds = (size_t)(SHV ? 2 : 4) * (BTV) * KSTEPS * 1024 + (EPKV == ST_FWD ? 0 : (size_t)4 * (16 + 64) * PB_RP * sizeof(bf16_t) + 4 * 128 * sizeof(float)); /* burst staging (+ z coefficients) */ \
    const long waves = (long)num_cus() * resident_per_cu(kern, 256, lds) * 4;                                           \
    const long wchunks = SHV ? (long)((nchunks + 3) / 4) * 4 : nchunks;   /* wave slots per row range */                \
    long tiles_per_item = (mtiles * wchunks + waves - 1) / waves;                                                       \
    if (tiles_per_item < 8) tiles_per_item = 8;                                                                         \
    if (tiles_per_item < min_tpi) tiles_per_item = min_tpi;                                                             \
    const long max_ranges = waves / wchunks > 0 ? waves / wchunks : 1;   /* one round of workgroups */ \
    if ((mtiles + tiles_per_item - 1) / tiles_per_item > max_ranges) tiles_per_item = (mtiles + max_ranges - 1) / max_ranges; \
    if (tiles_per_item > 4096) tiles_per_item = 4096;   /* keeps t * tile bytes in 32 bits (nt_st_kind) */              \
    const long nrg = (mtiles + tiles_per_item - 1) / tiles_per_item;                                                    \
    const long blocks = SHV ? nrg * ((nchunks + 3) / 4) : (nrg * nchunks + 3) / 4;                                      \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, A, W, ldw, ep, M, N, K, nchunks, (int)tiles_per_item, (float*)nullptr); \
  }
  // Bursts shared by the four waves of a workgroup (one read of the narrow operand per workgroup instead of per wave) where whole
  // groups of four chunks waste at most 5 % of the wave slots: 14x14 (23 / 27 chunks) -10..-18 %, 28x28 (12) equal; with 5 or 7 chunks
  // the repeated chunk costs more than the sharing saves (112x112 +38 %, 56x56 masked form +15 %; r03, tools/pwbench.py).
#define ST_CASE(EPKV)                                                                                                   \
  {                                                                                                                     \
    const bool shared = StCfg<KSTEPS, EPKV>::SH && ((nchunks + 3) / 4 * 4 - nchunks) * 20 <= nchunks;                   \
    if (shared) ST_LAUNCH(EPKV, (StCfg<KSTEPS, EPKV>::SH), (StCfg<KSTEPS, EPKV>::BT))                                   \
    else ST_LAUNCH(EPKV, false, (StCfg<KSTEPS, EPKV>::BTP))                                                            \
  }
  if (kind == ST_FWD) ST_CASE(ST_FWD) else ST_CASE(ST_MASK)
#undef ST_LAUNCH
#undef ST_CASE
}

static int launch_nt_ws(int mode, const Operand& A, const void* Wp, int ldw, const Epilogue& ep, long M, int N, int K, hipStream_t st) {
  const bf16_t* W = (const bf16_t*)Wp;
  const int wrows = (N + 63) / 64 * 64;   // rows of the packed weight matrix
  const int ncg = N > 64 ? 2 : 1;
  const int ngroups = (N + 64 * ncg - 1) / (64 * ncg);
  const bool do_stats = ep.stats && ep.stat_mode != STAT_NONE;
  const size_t lds_stat = do_stats ? (size_t)8 * 64 * ncg * sizeof(float) : 0;
  // two 16-row subtiles per wave (each weight fragment read feeds two MFMAs) when there are enough 128-row blocks
  constexpr int rt_env = 0;
  // measured in situ per shape (bs 256 step, same box, ATOMNAS_NT_WS_RT=1/2): one subtile per wave is faster wherever the prologue is
  // BN-apply (the projection forward: M = 50176 -28 %, 200704 -15 %, 802816 -7 % -- the two-subtile BNRELU instance with two chunks
  // needs 335 registers, one wave per SIMD) and on the small maps; two subtiles only pay with the two-stream BN-backward prologue
  // on the large maps (M = 200704: -3 %)
  const int rt = rt_env ? rt_env : ((mode == PRO_BNBWD && M >= 100000) ? 2 : 1);
#define WS_LAUNCH(MODE, NCGV, RTV)                                                                                       \
  {                                                                                                                      \
    auto kern = k_gemm_nt_ws<MODE, NCGV, RTV>;                                                                           \
    const size_t lds = (size_t)2 * 64 * NCGV * WS_WP * sizeof(bf16_t) + 2 * 3 * WS_KC * sizeof(float) + lds_stat;        \
    const long rblocks = (M + 64 * RTV - 1) / (64 * RTV);                                                                \
    long R = (long)num_cus() * resident_per_cu(kern, 256, lds) / ngroups;   /* row slots: one round of resident workgroups */ \
    if (R > rblocks) R = rblocks;                                                                                        \
    if (do_stats && R > ep.stat_rows) R = ep.stat_rows;                                                                  \
    if (R < 1) R = 1;                                                                                                    \
    hipLaunchKernelGGL(kern, dim3((unsigned)(R * ngroups)), dim3(256), lds, st, A, W, ldw, wrows, ep, M, N, K);          \
  }
#define WS_MODE(MODE)                                                                  \
  if (ncg == 1) { if (rt == 2) WS_LAUNCH(MODE, 1, 2) else WS_LAUNCH(MODE, 1, 1) }      \
  else { if (rt == 2) WS_LAUNCH(MODE, 2, 2) else WS_LAUNCH(MODE, 2, 1) }
  if (mode == PRO_NONE) { WS_MODE(PRO_NONE) }
  else if (mode == PRO_BNRELU) { WS_MODE(PRO_BNRELU) }
  else { WS_MODE(PRO_BNBWD) }
#undef WS_MODE
#undef WS_LAUNCH
  return check_launch("gemm_nt_ws");
}

// ------------------------------------------------------------------------------------------------ streaming wide-input GEMM
// C[M][N] = act(A * scale + shift) W^T for a NARROW output (N <= 48) of a wide slab-major input (K <= 448): the projection of the
// early stages (models/mobilenet_base.py:338,378), with the statistics of the output's BatchNorm.  The structure of k_expand_bwd_s:
// the weights are resident in LDS, A is copied HBM -> LDS by global_load_lds_dwordx4 in 64-row x 64-channel stages (8 contiguous 1 KB
// subtiles of the slab-major tensor) DEPTH - 1 stages ahead and waited for with a counted vmcnt; the B fragments are 16-byte LDS reads,
// the prologue runs on the fragment registers.  16 UT output channels per lane group instead of the 64 k_gemm_nt_ws always computes.
template <int UT, int NCH, int DEPTH>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_sw(const bf16_t* __restrict__ a, long ass, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int act, const bf16_t* __restrict__ W, int ldw,
                                                       bf16_t* __restrict__ cout, int ldc, float* __restrict__ stats, int stat_rows, long M,
                                                       int N, int K) {
  using T = bf16_t;
  using MM = Mma<T>;
  constexpr int KP = NCH * 64 + 8;
  constexpr int STAGE_B = 8 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_sw[];
  T* s_w = reinterpret_cast<T*>(smem_sw);                                                   // [16 UT][KP]
  float* s_co = reinterpret_cast<float*>(smem_sw + (size_t)16 * UT * KP * sizeof(T));      // [2][NCH * 64] scale, shift (zeros past K)
  float* s_stat = s_co + 2 * NCH * 64;                                                      // [4 waves][2][16 UT]
  unsigned char* s_st = reinterpret_cast<unsigned char*>(s_stat + 4 * 2 * 16 * UT);        // [DEPTH] stages

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int nchunk = (K + 63) / 64, nslabs = (K + 15) / 16;
  const long rblocks = (M + 63) / 64;
  const int rs = blockIdx.x, R = gridDim.x;
  const long nb = (rblocks - rs + R - 1) / R;

  for (int idx = tid; idx < 16 * UT * NCH * 8; idx += 256) {
    const int n = idx / (NCH * 8), k = (idx % (NCH * 8)) * 8;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)0.f;
    if (n < N && k < K) {
      o = *reinterpret_cast<const bf16x8*>(W + (long)n * ldw + k);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (k + e >= K) o[e] = (bf16_t)0.f;
    }
    *reinterpret_cast<bf16x8*>(s_w + n * KP + k) = o;
  }
  for (int i = tid; i < 2 * NCH * 64; i += 256) {
    const int v = i / (NCH * 64), k = i % (NCH * 64);
    s_co[i] = k < K ? (v ? shift[k] : scale[k]) : 0.f;
  }
  __syncthreads();

  const unsigned lds_st = (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)s_st);
  auto dma = [&](const T* g, unsigned dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(d) : "memory");
  };
  long ib = 0;
  int ic = 0, islot = 0;
  auto issue_next = [&]() {
    const bool live = ib < nb;
    const long rb = rs + (live ? ib : nb - 1) * R;
    const int c = live ? ic : nchunk - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // subtiles (row half i, channel tile wave) of the stage
      int slab = 4 * c + wave;
      slab = slab < nslabs ? slab : nslabs - 1;
      long row = rb * 64 + 32 * i + (lane >> 1);
      row = row < M ? row : M - 1;
      dma(a + slab * ass + row * 16 + 8 * (lane & 1), lds_st + (unsigned)islot * STAGE_B + (unsigned)(4 * i + wave) * 1024u);
    }
    ATOMNAS_RING_STAGE_END();
    islot = islot + 1 == DEPTH ? 0 : islot + 1;
    if (++ic == nchunk) { ic = 0; ++ib; }
  };

  const Act am = act_of(act);
  float ssum[UT][4], ssq[UT][4];
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[t][r] = ssq[t][r] = 0.f;

#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) issue_next();
  int slot = 0;
  const unsigned row_off = (unsigned)(((wave & 1) * 16 + j) * 32);
  for (long n = 0; n < nb; ++n) {
    const long rb = rs + n * R;
    const long row = rb * 64 + wave * 16 + j;
    f32x4 acc[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < nchunk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 2) : "memory");
        __syncthreads();
        issue_next();
        const unsigned base = lds_st + (unsigned)slot * STAGE_B;
        bf16x8 hb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned ad = base + (unsigned)(4 * (wave >> 1) + 2 * ks + (q >> 1)) * 1024u + row_off + (unsigned)(q & 1) * 16u;
          asm volatile("ds_read_b128 %0, %1" : "=v"(hb[ks]) : "v"(ad) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int kl = 64 * c + 32 * ks + 8 * q;
          float sc[8], sh[8], v[8];
          VecIO<float, 8>::load(s_co + kl, sc);
          VecIO<float, 8>::load(s_co + NCH * 64 + kl, sh);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)hb[ks][e] * sc[e] + sh[e];
          act_apply_v<8>(v, am);
          const bf16x8 af = MM::pack(v);
#pragma unroll
          for (int t = 0; t < UT; ++t) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(s_w + (16 * t + j) * KP + kl);
            acc[t] = MM::mma(wf, af, acc[t]);
          }
        }
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
      }
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const int c0 = 16 * t + 4 * q;
      if (row < M && c0 < N) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[t][r];
        u32x2 ob = __builtin_bit_cast(u32x2, o);
        asm volatile("" : "+v"(ob));
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const f32x2 f = bf16_pair_f32(ob[r2]);   // statistics see the stored value
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            ssum[t][2 * r2 + r] += f[r];
            ssq[t][2 * r2 + r] += f[r] * f[r];
          }
        }
        *reinterpret_cast<u32x2*>(cout + row * ldc + c0) = ob;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stats) {
    // rows of the wave (16 lanes j) by butterfly, the four waves in wave order; row rs of the partial-row buffer
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = ssum[t][r], s2 = ssq[t][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (j == 0) {
          s_stat[(wave * 2 + 0) * 16 * UT + 16 * t + 4 * q + r] = s1;
          s_stat[(wave * 2 + 1) * 16 * UT + 16 * t + 4 * q + r] = s2;
        }
      }
    __syncthreads();
    for (int i = tid; i < 2 * 16 * UT; i += 256) {
      const int pl = i / (16 * UT), c = i % (16 * UT);
      if (c < N) {
        float v = s_stat[(0 * 2 + pl) * 16 * UT + c];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += s_stat[(w * 2 + pl) * 16 * UT + c];
        const long elem = (long)pl * N + c;
        stats[(long)rs * 2 * N + elem] = v;
        stat_zero_tail(stats, 2L * N, rs + R, R, stat_rows, elem);
      }
    }
  }
}

// -1: not this kernel's case (the caller goes on), otherwise the launch status
static int launch_nt_sw(int mode, const Operand& A, const void* Wp, int ldw, const Epilogue& ep, long M, int N, int K, hipStream_t st) {
  static const int on = getenv("ATOMNAS_NT_SW") ? atoi(getenv("ATOMNAS_NT_SW")) : 1;   // experiment switch
  const bool do_stats = ep.stats && ep.stat_mode != STAT_NONE;
  if (!on || mode != PRO_BNRELU || A.ss1 <= 0 || N > 48 || N % 8 != 0 || K > 448 || K < 97 || M < 16384 || ep.out_f32 || ep.css != 0 ||
      ep.add || ep.z || ep.mask || ep.bias || (do_stats && ep.stat_mode != STAT_SQ))
    return -1;
  const int ut = (N + 15) / 16, nch = (K + 63) / 64 <= 5 ? 5 : 7;
  const size_t fixed = (size_t)16 * ut * (nch * 64 + 8) * sizeof(bf16_t) + (size_t)2 * nch * 64 * sizeof(float) + (size_t)4 * 2 * 16 * ut * sizeof(float);
  const int depth = 2 * (fixed + 4 * 8192) + 4096 <= max_lds_bytes() ? 4 : 3;
  const size_t lds = fixed + (size_t)depth * 8192;
  if (2 * lds + 2048 > max_lds_bytes()) return -1;
  const long rblocks = (M + 63) / 64;
#define SW_CASE(UTV, NCHV)                                                                                                              \
  if (ut == UTV && nch == NCHV) {                                                                                                       \
    auto kern = depth == 4 ? k_gemm_nt_sw<UTV, NCHV, 4> : k_gemm_nt_sw<UTV, NCHV, 3>;                                                   \
    long R = (long)num_cus() * resident_per_cu(kern, 256, lds);                                                                         \
    if (R > rblocks) R = rblocks;                                                                                                       \
    if (do_stats && R > ep.stat_rows) R = ep.stat_rows;                                                                                 \
    hipLaunchKernelGGL(kern, dim3((unsigned)R), dim3(256), lds, st, (const bf16_t*)A.p1, A.ss1, A.c1, A.c2, A.relu, (const bf16_t*)Wp, ldw, \
                       (bf16_t*)ep.c, ep.ldc, do_stats ? ep.stats : nullptr, ep.stat_rows, M, N, K);                                    \
    return check_launch("gemm_nt_sw");                                                                                                  \
  }
  SW_CASE(1, 5) SW_CASE(1, 7) SW_CASE(2, 5) SW_CASE(2, 7) SW_CASE(3, 5) SW_CASE(3, 7)
#undef SW_CASE
  return -1;
}

// ------------------------------------------------------------------------------------------------ streaming wide-input GEMM, late stages
// C[M][N] = act(A * scale + shift) W^T, statistics rows [sum c, sum c^2]: the projection forward of the 14x14 / 7x7 stages
// (models/mobilenet_base.py:338,378: N = 80 .. 320 output channels, K = 1440 .. 3456 hidden channels) on the LDS-DMA queue of
// k_gemm_nt_sw.  Difference: the weights (N x K x 2 bytes, up to 2.2 MB) are not resident.  A compiler-known weight load inside the loop
// would be waited for with a count that ignores the copies and drain the queue, so the 64-channel chunk of the
// weights AND the chunk's scale / shift travel through the same queue as the activations:
//   stage = 2 NWV activation subtiles [32 rows][16 channels] (1 KB each, contiguous in the slab-major tensor)
//         + 2 UT weight tiles of 8 rows x 64 channels (1 KB each; LDS row pitch 128 bytes, the 16-byte pieces of a row XOR-swizzled with
//           the row number -- the copy's LDS destination is lane-linear, so the swizzle is applied to the SOURCE address of each lane)
//         + 1 KB of coefficients (64 scale, 64 shift floats; every wave copies the same bytes: equal copy counts per wave).
// Everything else as in k_gemm_nt_sw: counted vmcnt, one barrier per stage, prologue on the fragment registers, 16 UT output channels.
// Round 4 prototype (profiles/r04_ntswg_prototype.txt: bit-identical to k_gemm_nt_ws, 13-20 % faster), in the library since round 5.
// NWV = 4: 64-row stages (two workgroups per CU where the LDS allows); NWV = 8: 128-row stages, one weight chunk per 128 rows (half the
// weight traffic from L2 per activation byte), one workgroup of eight waves per CU.
template <int UT, int DEPTH, int WGPC, int NWV>
__global__ __launch_bounds__(NWV * 64, WGPC) void k_gemm_nt_swg(const bf16_t* __restrict__ a, long ass, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, const bf16_t* __restrict__ W, int ldw,
                                                           bf16_t* __restrict__ cout, int ldc, float* __restrict__ stats, int stat_rows,
                                                           long M, int N, int K) {
  constexpr int NT = NWV * 64, RB = NWV * 16;   // threads, rows of a stage
  constexpr int AT = 2 * NWV;                   // activation subtiles [32 rows][16 channels] of a stage: (row group of 32, channel tile)
  constexpr int WT = 2 * UT;                    // weight tiles (8 rows x 64 channels) of a stage
  constexpr int WPW = (WT + NWV - 1) / NWV;     // ... per wave
  constexpr int CPS = 2 + WPW + 1;              // copies per wave and stage
  constexpr int STAGE_B = (AT + WT + 1) * 1024;  // bytes of a stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_swg[];
  float* s_stat = reinterpret_cast<float*>(smem_swg);                      // [NWV waves][2][16 UT]
  unsigned char* s_st = smem_swg + (size_t)NWV * 2 * 16 * UT * sizeof(float);   // [DEPTH] stages

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int nchunk = (K + 63) / 64, nslabs = (K + 15) / 16;
  const long rblocks = (M + RB - 1) / RB;
  const int rs = blockIdx.x, R = gridDim.x;
  const long nb = (rblocks - rs + R - 1) / R;

  const unsigned lds_st = (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)s_st);
  auto dma = [&](const void* g, unsigned dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(d) : "memory");
  };
  long ib = 0;
  int ic = 0, islot = 0;
  auto issue_next = [&]() {
    const bool live = ib < nb;
    const long rb = rs + (live ? ib : nb - 1) * R;
    const int c = live ? ic : nchunk - 1;
    const unsigned sb = lds_st + (unsigned)islot * STAGE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // activation subtiles wave, wave + NWV: (row group sub / 4, channel tile sub % 4)
      const int sub = wave + NWV * i;
      int slab = 4 * c + (sub & 3);
      slab = slab < nslabs ? slab : nslabs - 1;
      long row = rb * RB + 32 * (sub >> 2) + (lane >> 1);
      row = row < M ? row : M - 1;
      dma(a + slab * ass + row * 16 + 8 * (lane & 1), sb + (unsigned)sub * 1024u);
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {   // weight tiles wave, wave + NWV, ...: lane -> (row 8 tile + lane / 8, LDS piece lane % 8 <- source piece xor row)
      int wt = wave + NWV * i;
      wt = wt < WT ? wt : WT - 1;
      const int row = 8 * wt + (lane >> 3);
      int col = 64 * c + 8 * ((lane & 7) ^ (row & 7));
      col = col < ldw - 8 ? col : ldw - 8;   // past the packed pitch: any valid piece (meets activations zeroed by their coefficients)
      dma(W + (long)row * ldw + col, sb + (unsigned)(AT + wt) * 1024u);
    }
    {   // coefficients of the chunk: lanes 0..15 scale, 16..31 shift (32..63 repeat them), 4 floats each
      int k = 64 * c + 4 * (lane & 15);
      const int kmax = ((K + 3) & ~3) - 4;
      k = k < kmax ? k : kmax;
      dma(((lane >> 4) & 1 ? shift : scale) + k, sb + (unsigned)(AT + WT) * 1024u);
    }
    ATOMNAS_RING_STAGE_END();
    islot = islot + 1 == DEPTH ? 0 : islot + 1;
    if (++ic == nchunk) { ic = 0; ++ib; }
  };

  const Act am = act_of(act);
  float ssum[UT][4], ssq[UT][4];
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[t][r] = ssq[t][r] = 0.f;

#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) issue_next();
  int slot = 0;
  const unsigned row_off = (unsigned)(((wave & 1) * 16 + j) * 32);
  for (long n = 0; n < nb; ++n) {
    const long rb = rs + n * R;
    const long row = rb * RB + wave * 16 + j;
    f32x4 acc[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunk; ++c) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * CPS) : "memory");
      __syncthreads();
      issue_next();
      const unsigned base = lds_st + (unsigned)slot * STAGE_B;
      bf16x8 hb[2];
      f32x4 cf[2][4];   // [ks][scale lo, scale hi, shift lo, shift hi] of the lane's 8 channels
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned ad = base + (unsigned)(4 * (wave >> 1) + 2 * ks + (q >> 1)) * 1024u + row_off + (unsigned)(q & 1) * 16u;
        asm volatile("ds_read_b128 %0, %1" : "=v"(hb[ks]) : "v"(ad) : "memory");
        const unsigned cb = base + (unsigned)(AT + WT) * 1024u + (unsigned)(32 * ks + 8 * q) * 4u;
        asm volatile("ds_read_b128 %0, %1" : "=v"(cf[ks][0]) : "v"(cb) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(cf[ks][1]) : "v"(cb) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(cf[ks][2]) : "v"(cb) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:272" : "=v"(cf[ks][3]) : "v"(cb) : "memory");
      }
      // Round 6, the instances with one wave per SIMD (UT >= 12: the 7x7 maps, where a stage was a dependent chain read -> wait ->
      // arithmetic -> read -> wait -> MFMA, profiles/r06_late_stage_gemm_experiments.txt): the weight fragments of a k-step are in flight
      // while the prologue of the activations runs, and those of the second k-step while the MFMAs of the first issue.  LDS reads of a wave
      // return in order, so the waits are counted (lgkmcnt(LGK) = everything but the LGK fragment reads issued last; tools/check_asm_waits.py
      // models it).  Same MFMA order per accumulator: bit-identical results.  In situ: N192 K3456 0.251 -> 0.231 ms per step (three launches),
      // N320 0.111 -> 0.108; the 14x14 instances (two and more waves per SIMD overlap the chain by themselves) measured 1-2 % SLOWER with
      // it and keep the plain sequence.
      constexpr bool PREFETCH = UT >= 12;
      constexpr int LGK = UT < 15 ? UT : 15;   // the counter has four bits: with more fragment reads behind it the wait covers a few of them too
      bf16x8 wf[PREFETCH ? 2 : 1][UT];
      auto read_w = [&](int ks, int buf) {
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          const int wr = 16 * t + j, p = 4 * ks + q;
          const unsigned wa = base + (unsigned)AT * 1024u + (unsigned)wr * 128u + (unsigned)((p ^ (wr & 7)) << 4);
          asm volatile("ds_read_b128 %0, %1" : "=v"(wf[buf][t]) : "v"(wa) : "memory");
        }
      };
      if constexpr (PREFETCH) {
        read_w(0, 0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LGK) : "memory");   // activations and coefficients of both k-steps
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 af[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k0 = 64 * c + 32 * ks + 8 * q;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = cf[ks][e >> 2][e & 3], sh = cf[ks][2 + (e >> 2)][e & 3];
          v[e] = (float)hb[ks][e] * sc + sh;
        }
        act_apply_v<8>(v, am);
        // channels past K exist in the last chunk only (wave-uniform test): their copies re-read valid slabs and coefficients, so v is
        // finite there and is replaced by 0 after the activation (round 6: the per-element bound was 32 of the prologue's instructions
        // per k-step in every chunk)
        if (64 * c + 64 > K) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + e >= K) v[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) af[ks][e] = (bf16_t)v[e];
        if constexpr (!PREFETCH) {   // the k-step's weight fragments: all reads in flight, one wait, then its MFMAs
          read_w(ks, 0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][t], af[ks], acc[t], 0, 0, 0);
        }
      }
      if constexpr (PREFETCH) {
        __builtin_amdgcn_sched_barrier(0);
        read_w(1, 1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LGK) : "memory");   // the first k-step's fragments
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][t], af[0], acc[t], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][t], af[1], acc[t], 0, 0, 0);
      }
      slot = slot + 1 == DEPTH ? 0 : slot + 1;
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const int c0 = 16 * t + 4 * q;
      if (row < M && c0 < N) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[t][r];
        u32x2 ob = __builtin_bit_cast(u32x2, o);
        asm volatile("" : "+v"(ob));
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const f32x2 f = bf16_pair_f32(ob[r2]);   // statistics see the stored value
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            ssum[t][2 * r2 + r] += f[r];
            ssq[t][2 * r2 + r] += f[r] * f[r];
          }
        }
        *reinterpret_cast<u32x2*>(cout + row * ldc + c0) = ob;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stats) {
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = ssum[t][r], s2 = ssq[t][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (j == 0) {
          s_stat[(wave * 2 + 0) * 16 * UT + 16 * t + 4 * q + r] = s1;
          s_stat[(wave * 2 + 1) * 16 * UT + 16 * t + 4 * q + r] = s2;
        }
      }
    __syncthreads();
    for (int i = tid; i < 2 * 16 * UT; i += NT) {
      const int pl = i / (16 * UT), ch = i % (16 * UT);
      if (ch < N) {
        float v = s_stat[(0 * 2 + pl) * 16 * UT + ch];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v += s_stat[(w * 2 + pl) * 16 * UT + ch];
        const long elem = (long)pl * N + ch;
        stats[(long)rs * 2 * N + elem] = v;
        stat_zero_tail(stats, 2L * N, rs + R, R, stat_rows, elem);
      }
    }
  }
}

template <int UT, int WGPC, int NWV>
static int launch_swg(const bf16_t* a, long ass, const float* scale, const float* shift, int act, const bf16_t* W, int ldw, bf16_t* c, int ldc,
                      float* stats, int stat_rows, long M, int N, int K, hipStream_t st) {
  constexpr int DEPTH = 3;
  const size_t lds = (size_t)NWV * 2 * 16 * UT * sizeof(float) + (size_t)DEPTH * (2 * NWV + 2 * UT + 1) * 1024;
  if (lds > max_lds_bytes()) return -1;
  auto kern = k_gemm_nt_swg<UT, DEPTH, WGPC, NWV>;
  const long rblocks = (M + NWV * 16 - 1) / (NWV * 16);
  long R = (long)num_cus() * resident_per_cu(kern, NWV * 64, lds);
  if (R > rblocks) R = rblocks;
  if (stats && R > stat_rows) R = stat_rows;
  hipLaunchKernelGGL(kern, dim3((unsigned)R), dim3(NWV * 64), lds, st, a, ass, scale, shift, act, W, ldw, c, ldc, stats, stat_rows, M, N, K);
  return check_launch("gemm_nt_swg");
}


// -1: not this kernel's case (the caller continues with k_gemm_nt_ws)
static int launch_nt_swg(int mode, const Operand& A, const void* Wp, int ldw, const Epilogue& ep, long M, int N, int K, hipStream_t st) {
  static const int on = getenv("ATOMNAS_NT_SWG") ? atoi(getenv("ATOMNAS_NT_SWG")) : 1;   // experiment switch (0: k_gemm_nt_ws)
  const bool do_stats = ep.stats && ep.stat_mode != STAT_NONE;
  // measured (r04 prototype, r05 in situ): gains at 14 x 14 and 7 x 7 (M <= 50176), none at 28 x 28 (M = 200704: every stage moves
  // 12 KB of weights from L2 for 8 KB of activations) -- until round 6 took two thirds of the prologue's instructions out (common.h: Act):
  // since then 0.098 -> 0.077 ms at M = 200704, N = 40, K = 720 (tools/pwbench.py project), and the row limit is 262144
  static const long maxm = getenv("ATOMNAS_NT_SWG_MAXM") ? atol(getenv("ATOMNAS_NT_SWG_MAXM")) : 262144;   // experiment switch
  if (!on || mode != PRO_BNRELU || A.ss1 <= 0 || N % 8 != 0 || N > 320 || K < 256 || K % 4 != 0 || M < 8192 || M > maxm || ldw < 64 || ldw % 8 != 0 ||
      ep.out_f32 || ep.css != 0 || ep.add || ep.z || ep.mask || ep.bias || (do_stats && ep.stat_mode != STAT_SQ) || !A.c1 || !A.c2)
    return -1;
  const int ut = (N + 15) / 16;
  // 128-row stages (eight waves, one weight chunk per 128 rows: half the weight traffic per activation byte) where they still fill the
  // chip: 14 x 14 (392 workgroups); 64-row stages at 7 x 7 (196 workgroups of four waves)
  const bool wide = (M + 127) / 128 >= num_cus() && ut <= 6;
  float* stats = do_stats ? ep.stats : nullptr;
#define SWG_ARGS (const bf16_t*)A.p1, A.ss1, A.c1, A.c2, A.relu, (const bf16_t*)Wp, ldw, (bf16_t*)ep.c, ep.ldc, stats, ep.stat_rows, M, N, K, st
#define SWG_CASE(UTV, WG)                                   \
  if (ut == UTV) {                                          \
    if (wide) return launch_swg<UTV, 1, 8>(SWG_ARGS);       \
    return launch_swg<UTV, WG, 4>(SWG_ARGS);                \
  }
  SWG_CASE(3, 2) SWG_CASE(5, 2) SWG_CASE(6, 2)
  // wide outputs: four-wave stages only (an eight-wave instance would have 128 registers per lane for 2 x 16 UT accumulators and
  // weight fragments: it spills, and a scratch reload inside the ring loop drains the queue -- tools/check_asm_waits.py flags it)
  if (ut == 12) return launch_swg<12, 1, 4>(SWG_ARGS);
  if (ut == 20) return launch_swg<20, 1, 4>(SWG_ARGS);
#undef SWG_ARGS
#undef SWG_CASE
  return -1;
}

template <typename T>
static int launch_nt(int mode, const Operand& A, const void* Wp, int ldw, const Epilogue& ep, long M, int N, int K, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    int rc = launch_nt_sw(mode, A, Wp, ldw, ep, M, N, K, st);   // narrow output of a wide slab-major input: the streaming kernel
    if (rc >= 0) return rc;
    rc = launch_nt_swg(mode, A, Wp, ldw, ep, M, N, K, st);      // late stages: weights in the queue
    if (rc >= 0) return rc;
  }
  if constexpr (sizeof(T) == 2) {
    // column-stationary form when the output is the wide operand
    constexpr int cs_maxk = 192;
    // with the BatchNorm-backward prologue the 6-k-step instance (K = 192: 7x7 maps) needs 256 + 49 registers, one wave per SIMD;
    // the LDS-weights kernel is 10 % faster there (in situ, M = 12544, N = 3456 / 1728), without a prologue it is 50 % slower
    constexpr int cs_maxk_pro = 96;
    if (K <= (mode == PRO_BNBWD ? cs_maxk_pro : cs_maxk) && K <= 192 && N >= 2 * K && N >= 96 && M >= 1024) {
      const int ksteps = (K + 31) / 32;
      if (const int kind = nt_st_kind(mode, A, ep, M, N, K)) {
        if (ksteps == 1) launch_nt_st<1>(kind, A, Wp, ldw, ep, M, N, K, st);
        else if (ksteps == 2) launch_nt_st<2>(kind, A, Wp, ldw, ep, M, N, K, st);
        else if (ksteps == 3) launch_nt_st<3>(kind, A, Wp, ldw, ep, M, N, K, st);
        else launch_nt_st<6>(kind, A, Wp, ldw, ep, M, N, K, st);
        return check_launch("gemm_nt_st");
      }
      // anything else in this shape class (a prologue, a bias / residual epilogue, K not a multiple of 8) takes the kernels below
    }
  }
  if constexpr (sizeof(T) == 2) {
    constexpr int small_env = 1;
    if (small_env && N <= 64 && K <= 64 && M >= 65536) {
      const long mtiles = (M + 15) / 16;
      const bool do_stats = ep.stats && ep.stat_mode != STAT_NONE;
      const size_t lds = do_stats ? (size_t)8 * 64 * sizeof(float) : 0;
      const bf16_t* W = (const bf16_t*)Wp;
      // lanes own 4 channels x NT tiles where the output allows it (bf16, N a multiple of 4); ATOMNAS_NT_SMALL_NARROW=0: the 16-channel form
      static const int narrow_on = getenv("ATOMNAS_NT_SMALL_NARROW") ? atoi(getenv("ATOMNAS_NT_SMALL_NARROW")) : 1;
      const int ntile = (narrow_on && !ep.out_f32 && N % 4 == 0) ? (N <= 16 ? 1 : N <= 32 ? 2 : 4) : 0;
#define SM_LAUNCH(MODE, KSTV)                                                                                  \
  {                                                                                                            \
    auto kern = ntile == 1 ? k_gemm_nt_small<MODE, KSTV, 1> : ntile == 2 ? k_gemm_nt_small<MODE, KSTV, 2>      \
              : ntile == 4 ? k_gemm_nt_small<MODE, KSTV, 4> : k_gemm_nt_small<MODE, KSTV, 0>;                  \
    long R = (long)num_cus() * resident_per_cu(kern, 256, lds);                                                \
    if (R > (mtiles + 3) / 4) R = (mtiles + 3) / 4;                                                            \
    if (do_stats && R > ep.stat_rows) R = ep.stat_rows;                                                        \
    if (R < 1) R = 1;                                                                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)R), dim3(256), lds, st, A, W, ldw, ep, M, N, K);                   \
  }
#define SM_CASE(MODE) \
  if (K <= 32) SM_LAUNCH(MODE, 1) else SM_LAUNCH(MODE, 2)
      if (mode == PRO_NONE) { SM_CASE(PRO_NONE) }
      else if (mode == PRO_BNRELU) { SM_CASE(PRO_BNRELU) }
      else { SM_CASE(PRO_BNBWD) }
#undef SM_CASE
#undef SM_LAUNCH
      return check_launch("gemm_nt_small");
    }
  }
  if constexpr (sizeof(T) == 2) {
    constexpr int ws_env = 1;
    constexpr int ws_mink = 97;
    if (ws_env && K >= ws_mink && M >= 4096) return launch_nt_ws(mode, A, Wp, ldw, ep, M, N, K, st);
  }
  constexpr int KS = 4 * Mma<T>::EPL;
  const int Kpad = (K + KS - 1) / KS * KS;
  const long mtiles = (M + 15) / 16;
  const bool wide = N > 64;  // keep two 64-channel chunks live when there is more than one
  const int ngroups = wide ? (N + 127) / 128 : 1;
  const bool do_stats = ep.stats && ep.stat_mode != STAT_NONE;
  const size_t lds = do_stats ? (size_t)8 * (wide ? 128 : 64) * sizeof(float) : 0;
  const T* W = (const T*)Wp;
  // persistent grid-stride loop over (row tile, channel group) items: one round of resident workgroups
#define NT_LAUNCH(MODE, NCGV)                                                                                  \
  {                                                                                                            \
    auto kern = k_gemm_nt<T, MODE, NCGV>;                                                                      \
    long R = (long)num_cus() * resident_per_cu(kern, 256, lds) / ngroups;   /* row slots of 4 tiles */         \
    if (R > (mtiles + 3) / 4) R = (mtiles + 3) / 4;                                                            \
    if (do_stats && R > ep.stat_rows) R = ep.stat_rows;                                                        \
    if (R < 1) R = 1;                                                                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)(R * ngroups)), dim3(256), lds, st, A, W, ldw, ep, M, N, K, Kpad); \
  }
#define NT_CASE(MODE) \
  if (wide) NT_LAUNCH(MODE, 2) else NT_LAUNCH(MODE, 1)
  if (mode == PRO_NONE) { NT_CASE(PRO_NONE) }
  else if (mode == PRO_BNRELU) { NT_CASE(PRO_BNRELU) }
  else { NT_CASE(PRO_BNBWD) }
#undef NT_CASE
#undef NT_LAUNCH
  return check_launch("gemm_nt");
}

// fused project backward (column-stationary input gradient + weight gradient)
// the streaming form of the fused projection backward: A is the materialised dP (atomnas_bnbwd_apply), no prologue
template <int KSTEPS, int UT>
static int launch_project_bwd_st(const Operand& A, const bf16_t* W, int ldw, const Epilogue& ep, float* dwp, long si, long sj, float* ws,
                                 long ws_floats, long M, int N, int K, hipStream_t st) {
  using Cfg = StCfg<KSTEPS, ST_PBWD>;
  const int nchunks = (N + 63) / 64;
  const long mtiles = (M + 15) / 16;
  auto kern = k_gemm_nt_st<KSTEPS, ST_PBWD, Cfg::PD, Cfg::BT, Cfg::WPE, UT, false>;
  const size_t lds = (size_t)4 * Cfg::BT * KSTEPS * 1024 + (size_t)4 * (16 * UT + 64) * PB_RP * sizeof(bf16_t) + 4 * 128 * sizeof(float);
  const long waves = (long)num_cus() * resident_per_cu(kern, 256, lds) * 4;
  long tiles_per_item = (mtiles * nchunks + waves - 1) / waves;
  if (tiles_per_item < 8) tiles_per_item = 8;
  const long min_tpi = (mtiles + ep.stat_rows - 1) / ep.stat_rows;   // every row range owns one partial row of the statistics
  if (tiles_per_item < min_tpi) tiles_per_item = min_tpi;
  const long max_parts = ws_floats / ((long)K * N);   // ... and one partial of the weight gradient
  ATOMNAS_REQUIRE(max_parts >= 1, "project_bwd: workspace too small for one partial (%ld floats)", (long)K * N);
  long max_ranges = waves / nchunks > 0 ? waves / nchunks : 1;
  if (max_ranges > max_parts) max_ranges = max_parts;
  if ((mtiles + tiles_per_item - 1) / tiles_per_item > max_ranges) tiles_per_item = (mtiles + max_ranges - 1) / max_ranges;
  ATOMNAS_REQUIRE(tiles_per_item <= 4096, "project_bwd: %ld row tiles per work item (workspace / statistics rows too small for M=%ld)", tiles_per_item, M);
  const long nranges = (mtiles + tiles_per_item - 1) / tiles_per_item;
  const long items = nranges * nchunks;
  hipLaunchKernelGGL(kern, dim3((unsigned)((items + 3) / 4)), dim3(256), lds, st, A, W, ldw, ep, M, N, K, nchunks, (int)tiles_per_item, ws);
  if (int rc = check_launch("project_bwd_st")) return rc;
  return reduce_parts(ws, (long)K * N, (int)nranges, (long)K * N, dwp, N, si, sj, st);
}

// ------------------------------------------------------------------------------------------------ streaming expand backward (e == NULL)
// The one-stream form of the expand backward (dE = c1 * h, atomnas_expand_bwd with e == NULL) as a pure stream of h:
//   gx[m][n]  = sum_k h[m][k] * (c1[k] We[k][n])  (+ x M + v + add)          -- c1 folded into the RESIDENT weights, once per workgroup
//   R[n][k]   = c1[k] * sum_m x[m][n] h[m][k]                                 -- c1 applied to the accumulators, once per workgroup
// so the loop body has no prologue arithmetic at all: a stage (64 rows x 64 hidden channels of h = 8 row-major [32][16] subtiles, each
// 1 KB CONTIGUOUS in the slab-major tensor) is copied HBM -> LDS by global_load_lds_dwordx4 into a ring of DEPTH stages, DEPTH - 1
// stages ahead of the MFMAs, and waited for with a COUNTED vmcnt (the structure of k_gemm_tn3).  The fragments of gx's product are
// plain 16-byte LDS reads of those subtiles, the k-major fragments of the weight-gradient product come out of the same bytes with
// ds_read_b64_tr_b16.  The block's x rows (and the residual rows) ride in the same queue, one small tile per row block.
// k_expand_bwd kept ONE 64-channel chunk of h per wave in flight in registers behind conditional loads: 16 KB per CU in flight, which at
// ~2 us of loaded HBM latency is the 2.0 TB/s it measured (Little's law), whatever its instruction count.  Here: 48 KB per CU.
// Every vector-memory operation of the loop is issued by this code (the compiler knows of none but the epilogue's stores, which need
// no wait), so the only vmcnt waits are the counted ones.
template <int UT, int NCH, int DEPTH>
__global__ __launch_bounds__(256, 2) void k_expand_bwd_s(const bf16_t* __restrict__ h, long hss, const float* __restrict__ c1,
                                                         const bf16_t* __restrict__ Wt, int ldw, const bf16_t* __restrict__ x, int ldx,
                                                         const bf16_t* __restrict__ add, int ldadd, bf16_t* __restrict__ gx, int ldgx,
                                                         const bf16_t* __restrict__ mpk, int ldm, const float* __restrict__ vb,
                                                         float* __restrict__ ws, long M, int N, int K) {
  using T = bf16_t;
  using MM = Mma<T>;
  constexpr int KP = NCH * 64 + 8;   // pitch of the resident weights (elements): 16-byte aligned rows, 4 banks apart
  constexpr int STAGE_B = 8 * 1024;  // bytes of a stage: 8 subtiles [32 rows][16 channels]
  constexpr int XT_B = 2 * UT * 1024;   // bytes of an x (or residual) tile: 2 row halves x UT channel tiles
  constexpr int XPASS = (2 * UT + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_xs[];
  T* s_w = reinterpret_cast<T*>(smem_xs);                                  // [16 UT][KP]  c1[k] * We[k][n], row n
  unsigned char* s_st = smem_xs + (size_t)16 * UT * KP * sizeof(T);       // [DEPTH] stages
  unsigned char* s_x = s_st + DEPTH * STAGE_B;                             // [2] x tiles
  unsigned char* s_a = s_x + 2 * XT_B;                                     // [2] residual tiles

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int nchunk = (K + 63) / 64, nslabs = (K + 15) / 16;
  const long rblocks = (M + 63) / 64;
  const int rs = blockIdx.x, R = gridDim.x;
  const long nb = (rblocks - rs + R - 1) / R;   // row blocks of this workgroup (rs < rblocks)

  // resident weights, scaled by c1 (zero rows / columns past N / K: whatever a clamped copy brings in there meets zeros)
  for (int idx = tid; idx < 16 * UT * NCH * 8; idx += 256) {
    const int n = idx / (NCH * 8), k = (idx % (NCH * 8)) * 8;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)0.f;
    if (n < N && k < K) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(Wt + (long)n * ldw + k);
      float c[8];
      VecIO<float, 8>::load(c1 + k, c);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((k + e < K) ? (float)w[e] * c[e] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(s_w + n * KP + k) = o;
  }
  // x M correction (A fragments of M, rows = output channels) and the bias, in registers for the whole kernel
  constexpr int MKS = (UT + 1) / 2;   // k-steps (32 x channels each) of the x M product
  bf16x8 mf[MKS][UT];
  float vbv[UT][4];
#pragma unroll
  for (int t = 0; t < UT; ++t) {
#pragma unroll
    for (int ks = 0; ks < MKS; ++ks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) mf[ks][t][e] = (bf16_t)0.f;
      if (mpk && 32 * ks + 8 * q < ldm) mf[ks][t] = *reinterpret_cast<const bf16x8*>(mpk + (long)(16 * t + j) * ldm + 32 * ks + 8 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) vbv[t][r] = (vb && 16 * t + 4 * q + r < N) ? vb[16 * t + 4 * q + r] : 0.f;
  }
  __syncthreads();   // (also: the loads above are complete before the counted waits below start counting)

  const unsigned lds_st = (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)s_st);
  const unsigned lds_x = (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)s_x);
  const unsigned lds_a = (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)s_a);
  auto dma = [&](const T* g, unsigned dst) {   // 64 lanes x 16 bytes -> LDS bytes dst .. dst + 1023, lane-linear
    const unsigned d = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(d) : "memory");
  };
  // issue side of the queue: item = (row block ib of this workgroup, chunk ic); past the end the last item is copied again
  long ib = 0;
  int ic = 0, islot = 0;
  auto issue_next = [&]() {
    const bool live = ib < nb;
    const long rb = rs + (live ? ib : nb - 1) * R;
    const int c = live ? ic : nchunk - 1;
    if (live && ic == 0) {   // the row block's x rows (and residual rows): sub = (row half, channel tile), lane -> (row, 8-channel half)
#pragma unroll
      for (int i = 0; i < XPASS; ++i) {
        int sub = wave + 4 * i;
        sub = sub < 2 * UT ? sub : 2 * UT - 1;
        const int rh = sub / UT, ct = sub % UT;
        long row = rb * 64 + rh * 32 + (lane >> 1);
        row = row < M ? row : M - 1;
        const int ch = 16 * ct + 8 * (lane & 1);
        dma(x + row * ldx + (ch < ldx - 8 ? ch : ldx - 8), lds_x + (unsigned)(ib & 1) * XT_B + (unsigned)sub * 1024u);
        if (add) dma(add + row * ldadd + (ch < ldadd - 8 ? ch : ldadd - 8), lds_a + (unsigned)(ib & 1) * XT_B + (unsigned)sub * 1024u);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // subtiles wave, wave + 4 of the stage: (row half, channel tile) = (i, wave)
      int slab = 4 * c + wave;
      slab = slab < nslabs ? slab : nslabs - 1;
      long row = rb * 64 + 32 * i + (lane >> 1);
      row = row < M ? row : M - 1;
      dma(h + slab * hss + row * 16 + 8 * (lane & 1), lds_st + (unsigned)islot * STAGE_B + (unsigned)(4 * i + wave) * 1024u);
    }
    ATOMNAS_RING_STAGE_END();
    islot = islot + 1 == DEPTH ? 0 : islot + 1;
    if (++ic == nchunk) { ic = 0; ++ib; }
  };

  const unsigned tr_lane = (unsigned)(((8 * q + (j >> 2)) * 16 + 4 * (j & 3)) * 2);
  auto tr_frag = [&](unsigned sub_addr) {
    const unsigned a0 = sub_addr + tr_lane;
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:128" : "=v"(hi) : "v"(a0) : "memory");
    return TrFrag{lo, hi};
  };

  f32x4 racc[NCH][UT];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < UT; ++t) racc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) issue_next();
  int slot = 0;
  // this wave's rows of a block: 16 wave + j, i.e. row half wave / 2, row (wave % 2) * 16 + j of the half
  const unsigned row_off = (unsigned)(((wave & 1) * 16 + j) * 32);
  for (long n = 0; n < nb; ++n) {
    const long rb = rs + n * R;
    const long row = rb * 64 + wave * 16 + j;
    f32x4 acc[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xa[2][UT];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < nchunk) {
        // stage (n, c) has landed when at most the copies of the DEPTH - 2 later stages are outstanding (anything else issued since
        // only makes the wait stricter); the barrier extends that to the other waves' copies and frees the slot of the stage before
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 2) : "memory");
        __syncthreads();
        issue_next();
        const unsigned base = lds_st + (unsigned)slot * STAGE_B;
        if (c == 0) {   // x^T fragments of the row block (rows past M cut off: their copies read row M - 1)
          const unsigned xb = lds_x + (unsigned)(n & 1) * XT_B;
          TrFrag xf[2][UT];
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int t = 0; t < UT; ++t) xf[k2][t] = tr_frag(xb + (unsigned)(k2 * UT + t) * 1024u);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const long rows_left = M - (rb * 64 + 32 * k2);
#pragma unroll
            for (int t = 0; t < UT; ++t) {
              xa[k2][t] = __builtin_shufflevector(xf[k2][t].lo, xf[k2][t].hi, 0, 1, 2, 3, 4, 5, 6, 7);
              if (rows_left < 32) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                  if (8 * q + e >= rows_left) xa[k2][t][e] = (bf16_t)0.f;
              }
            }
          }
        }
        TrFrag hf[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) hf[k2] = tr_frag(base + (unsigned)(4 * k2 + wave) * 1024u);
        bf16x8 hb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned a = base + (unsigned)(4 * (wave >> 1) + 2 * ks + (q >> 1)) * 1024u + row_off + (unsigned)(q & 1) * 16u;
          asm volatile("ds_read_b128 %0, %1" : "=v"(hb[ks]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < UT; ++t) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(s_w + (16 * t + j) * KP + 64 * c + 32 * ks + 8 * q);
            acc[t] = MM::mma(wf, hb[ks], acc[t]);
          }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bf16x8 hv = __builtin_shufflevector(hf[k2].lo, hf[k2].hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int t = 0; t < UT; ++t) racc[c][t] = MM::mma(xa[k2][t], hv, racc[c][t]);
        }
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
      }
    }
    // + x M (the c2 term of the BatchNorm backward), bias, residual; rows past M and channels past N are not stored
    const unsigned xrow = lds_x + (unsigned)(n & 1) * XT_B + (unsigned)((wave >> 1) * UT) * 1024u + row_off;
    if (mpk) {
#pragma unroll
      for (int ks = 0; ks < MKS; ++ks) {
        bf16x8 xv;
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = (bf16_t)0.f;
        const int ct = 2 * ks + (q >> 1);   // the lane's 8 x channels 32 ks + 8 q ..: tile ct, half q & 1
        if (ct < UT) {
          const unsigned a = xrow + (unsigned)ct * 1024u + (unsigned)(q & 1) * 16u;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(xv) : "v"(a) : "memory");
        }
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = MM::mma(mf[ks][t], xv, acc[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const int c0 = 16 * t + 4 * q;
      bf16x4 av;
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = (bf16_t)0.f;
      if (add) {
        const unsigned a = lds_a + (unsigned)(n & 1) * XT_B + (unsigned)((wave >> 1) * UT + t) * 1024u + row_off + (unsigned)(8 * q);
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(av) : "v"(a) : "memory");
      }
      if (row < M && c0 < N) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[t][r] + vbv[t][r] + (float)av[r]);
        *reinterpret_cast<bf16x4*>(gx + row * ldgx + c0) = o;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus copies of the last stages must not outlive the workgroup's LDS

  // this workgroup's partial of R = c1 * x^T h: element (x channel uc, hidden channel vc) at ws[(rs * N + uc) * K + vc]
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int vc = c * 64 + 16 * wave + j;
    if (c < nchunk && vc < K) {
      const float cv = c1[vc];
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int uc = 16 * t + 4 * q + r;
          if (uc < N) ws[((long)rs * N + uc) * K + vc] = racc[c][t][r] * cv;
        }
    }
  }
}

template <int UT, int NCH>
static int launch_expand_bwd_s(const bf16_t* h, long hss, const float* c1, const bf16_t* W, int ldw, const bf16_t* x, int ldx,
                               const bf16_t* add, int ldadd, bf16_t* gx, int ldgx, const bf16_t* mpk, int ldm, const float* vb, float* dwe,
                               float* ws, long ws_floats, long M, int N, int K, hipStream_t st) {
  // The x / residual tiles are double-buffered per row block: the copy for block b + 2 is issued DEPTH - 1 stages ahead of that block's
  // first stage, which lies inside block b + 1 only if a block has at least DEPTH - 1 = 3 stages.  Narrower hidden tensors (K <= 128:
  // one or two 64-channel chunks) take k_expand_bwd.
  if ((K + 63) / 64 < 3) return -1;
  const size_t fixed = (size_t)16 * UT * (NCH * 64 + 8) * sizeof(bf16_t) + (size_t)4 * 2 * UT * 1024;   // weights + x and residual tiles
  const int depth = 2 * (fixed + 4 * 8192) + 4096 <= max_lds_bytes() ? 4 : 3;   // deepest ring that leaves room for two workgroups per CU
  const size_t lds = fixed + (size_t)depth * 8192;
  if (lds > max_lds_bytes()) return -1;
  auto kern = depth == 4 ? k_expand_bwd_s<UT, NCH, 4> : k_expand_bwd_s<UT, NCH, 3>;
  const long rblocks = (M + 63) / 64;
  long R = (long)num_cus() * resident_per_cu(kern, 256, lds);   // one round of resident workgroups
  if (R > rblocks) R = rblocks;
  const long max_parts = ws_floats / ((long)N * K);   // every workgroup owns one partial of the weight gradient
  if (R > max_parts) R = max_parts;
  ATOMNAS_REQUIRE(R >= 1, "expand_bwd: workspace too small for one partial (%ld floats)", (long)N * K);
  hipLaunchKernelGGL(kern, dim3((unsigned)R), dim3(256), lds, st, h, hss, c1, W, ldw, x, ldx, add, ldadd, gx, ldgx, mpk, ldm, vb, ws, M, N, K);
  if (int rc = check_launch("expand_bwd(stream)")) return rc;
  return reduce_parts(ws, (long)N * K, (int)R, (long)N * K, dwe, K, 1, N, st);
}

// fused expand backward: supported shapes and launch
static inline int xb_nch(int HT) {
  const int n = (HT + WS_KC - 1) / WS_KC;
  return n <= 5 ? 5 : (n <= 7 ? 7 : (n <= 12 ? 12 : 0));
}
template <int UT, int NCH>
static int launch_expand_bwd_cfg(const Operand& A, const bf16_t* W, int ldw, int wrows, const bf16_t* x, int ldx, const Epilogue& ep, float* dwe,
                                 float* ws, long ws_floats, long M, int N, int K, const bf16_t* mpk, int ldm, hipStream_t st) {
  auto kern = k_expand_bwd<UT, NCH>;
  const size_t lds = (size_t)2 * 64 * WS_WP * sizeof(bf16_t) + 2 * 3 * WS_KC * sizeof(float) + (size_t)2 * 64 * XB_DP * sizeof(bf16_t) +
                     (size_t)16 * UT * XB_DP * sizeof(bf16_t);
  const long rblocks = (M + 63) / 64;
  long R = (long)num_cus() * resident_per_cu(kern, 256, lds);   // one round of resident workgroups
  if (R > rblocks) R = rblocks;
  const long max_parts = ws_floats / ((long)N * K);   // every workgroup owns one partial of the weight gradient
  if (R > max_parts) R = max_parts;
  ATOMNAS_REQUIRE(R >= 1, "expand_bwd: workspace too small for one partial (%ld floats)", (long)N * K);
  hipLaunchKernelGGL(kern, dim3((unsigned)R), dim3(256), lds, st, A, W, ldw, wrows, x, ldx, ep, ws, M, N, K, mpk, ldm);
  if (int rc = check_launch("expand_bwd")) return rc;
  // dWe[n * inp + k] += sum over workgroups of R[k][n], in workgroup order
  return reduce_parts(ws, (long)N * K, (int)R, (long)N * K, dwe, K, 1, N, st);
}

}  // namespace atomnas

#if WS_TIMING
extern "C" int atomnas_debug_ws_timing(unsigned long long* out8, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(atomnas::g_ws_timing), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(atomnas::g_ws_timing), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
using namespace atomnas;

// C[M,N] = epilogue( prologue(A)[M,K] x Wp[N,K]^T ).  Wp is the packed weight (storage dtype, row pitch ldw >= K rounded
// up to the MFMA k-step, rows padded to a multiple of 64, padding zero) produced by atomnas_pack_weights.
extern "C" int atomnas_pw_gemm_nt(int a_mode, const void* a, int lda, long a_ss, const void* a2, int lda2, long a2_ss, const float* ac1,
                                  const float* ac2, const float* ac3, int a_relu, const void* wp, int ldw, void* c, int ldc, long c_ss,
                                  int out_f32, const void* add, int ldadd, const void* z, int ldz, long z_ss, const float* zscale,
                                  const float* zshift,
                                  int mask, const float* bias, float* stats, int stat_mode, int stat_rows, long M, int N, int K,
                                  int dtype, void* stream) {
  ATOMNAS_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "pw_gemm_nt: bad dtype %d", dtype);
  ATOMNAS_REQUIRE(a_mode >= PRO_NONE && a_mode <= PRO_BNBWD, "pw_gemm_nt: bad prologue %d", a_mode);
  ATOMNAS_REQUIRE(M > 0 && N > 0 && K > 0, "pw_gemm_nt: empty shape");
  ATOMNAS_REQUIRE(wp && c && (c_ss >= M * 16 || (c_ss == 0 && ldc >= N && ldc % 8 == 0)), "pw_gemm_nt: bad output/weights");
  ATOMNAS_REQUIRE(a_ss == 0 || a_ss >= M * 16, "pw_gemm_nt: slab stride of A smaller than M*16");
  {
    const int ks = (dtype == DT_BF16) ? 32 : 4;
    ATOMNAS_REQUIRE(ldw >= (K + ks - 1) / ks * ks && ldw % 8 == 0, "pw_gemm_nt: packed weight pitch %d too small for K=%d", ldw, K);
  }
  ATOMNAS_REQUIRE(!stats || N <= NT_MAX_STAT, "pw_gemm_nt: N=%d too wide for fused statistics", N);
  ATOMNAS_REQUIRE(!stats || stat_mode == STAT_NONE || stat_rows > 0, "pw_gemm_nt: statistics need stat_rows > 0");
  ATOMNAS_REQUIRE(!add || (ldadd >= N && ldadd % 8 == 0), "pw_gemm_nt: bad residual pitch");
  ATOMNAS_REQUIRE(!z || z_ss >= M * 16 || (z_ss == 0 && ldz >= N && ldz % 8 == 0), "pw_gemm_nt: bad z pitch");
  ATOMNAS_REQUIRE(!mask || (z && zscale && zshift), "pw_gemm_nt: mask needs z, zscale, zshift");
  ATOMNAS_REQUIRE(stat_mode != STAT_Z || z, "pw_gemm_nt: STAT_Z needs z");
  Operand A{a, lda, a2, lda2, a_ss, a2_ss, ac1, ac2, ac3, a_relu};
  if (check_operand("pw_gemm_nt", A, a_mode, K)) return 1;
  Epilogue ep{c, ldc, out_f32, add, ldadd, z, ldz, c_ss, z_ss, zscale, zshift, mask, bias, stats, stat_mode, stat_rows};
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32) return launch_nt<float>(a_mode, A, wp, ldw, ep, M, N, K, st);
  return launch_nt<bf16_t>(a_mode, A, wp, ldw, ep, M, N, K, st);
}

// 1 when atomnas_expand_bwd has an instance for this shape: bf16 storage, and the weight-gradient accumulators (4 registers per
// (16 input channels, 64 hidden channels) tile) leave room for two workgroups per CU.  Measured in situ (bs 256): 24 -> 432 and
// 16 -> 288 run 1.7x / 1.4x faster than the two GEMMs; 40 -> 720 (144 accumulator registers, one workgroup per CU, spills) was
// 1.7x SLOWER and has no instance.
extern "C" int atomnas_expand_bwd_supported(int inp, int hid, int dtype) {
  if (dtype != DT_BF16 || inp < 1 || inp > 48 || hid < 1 || xb_nch(hid) == 0) return 0;
  return 4 * ((inp + 15) / 16) * xb_nch(hid) <= XB_MINB2_LIMIT;
}

// Backward of the expand convolution (models/mobilenet_base.py:316-320) without its raw output E, both gradients from ONE pass over h:
//   dE = c1*h (the c2*E + c3 part of the BatchNorm backward: atomnas_xb_coeffs);  gx[M, inp] = dE * We (+ x M + v) (+ add);
//   dwe[n * inp + k] += sum_m dE[m][n] * x[m][k].   wt: We^T packed by atomnas_pack_weights ([inp padded to 64][ldw >= hid]).
extern "C" int atomnas_expand_bwd(const void* h, int ldh, long h_ss, const float* c1, const void* x, int ldx, const void* wt, int ldw,
                                  const void* add, int ldadd, void* gx, int ldgx, float* dwe, float* ws, long ws_floats, const void* mp, int ldm,
                                  const float* vb, long M, int inp, int hid, int dtype, void* stream) {
  ATOMNAS_REQUIRE(atomnas_expand_bwd_supported(inp, hid, dtype), "expand_bwd: unsupported shape inp=%d hid=%d dtype=%d", inp, hid, dtype);
  ATOMNAS_REQUIRE(h && c1 && x && wt && gx && dwe && ws && M > 0, "expand_bwd: bad arguments");
  ATOMNAS_REQUIRE(h_ss >= M * 16 || (h_ss == 0 && ldh >= hid && ldh % 8 == 0), "expand_bwd: bad hidden layout");
  ATOMNAS_REQUIRE(!mp || (ldm >= (inp + 31) / 32 * 32 && ldm % 8 == 0), "expand_bwd: bad pitch of the x M term (ldm=%d)", ldm);
  ATOMNAS_REQUIRE(ldx >= inp && ldx % 8 == 0 && ldgx >= inp && ldgx % 8 == 0 && (!add || (ldadd >= inp && ldadd % 8 == 0)), "expand_bwd: bad pitch");
  ATOMNAS_REQUIRE(ldw >= (hid + 31) / 32 * 32 && ldw % 8 == 0, "expand_bwd: packed weight pitch %d too small for hid=%d", ldw, hid);
  const float *c2 = c1, *c3 = c1;   // the register-prefetch kernel stages three coefficient vectors per chunk; only the first is used
  Operand A{h, ldh, nullptr, 0, h_ss, 0, c1, c2, c3, 0};
  Epilogue ep{gx, ldgx, 0, add, ldadd, nullptr, 0, 0, 0, nullptr, nullptr, 0, mp ? vb : nullptr, nullptr, STAT_NONE, 0};
  hipStream_t st = (hipStream_t)stream;
  const int ut = (inp + 15) / 16, nch = xb_nch(hid);
  const bf16_t* W = (const bf16_t*)wt;
  const bf16_t* X = (const bf16_t*)x;
  const int wrows = (inp + 63) / 64 * 64;
  // slab-major h: the streaming kernel (ATOMNAS_XB_STREAM=0: experiment switch back to k_expand_bwd)
  static const int xs_on = getenv("ATOMNAS_XB_STREAM") ? atoi(getenv("ATOMNAS_XB_STREAM")) : 1;
  if (xs_on && h_ss > 0 && M >= 64 && ldx >= 8 && (!add || ldadd >= 8)) {
#define XS_CASE(UTV, NCHV)                                                                                                             \
  if (ut == UTV && nch == NCHV) {                                                                                                      \
    const int rc = launch_expand_bwd_s<UTV, NCHV>((const bf16_t*)h, h_ss, c1, W, ldw, X, ldx, (const bf16_t*)add, ldadd, (bf16_t*)gx, ldgx, \
                                                  (const bf16_t*)mp, ldm, mp ? vb : nullptr, dwe, ws, ws_floats, M, inp, hid, st);     \
    if (rc >= 0) return rc;                                                                                                            \
  }
    XS_CASE(1, 5) XS_CASE(1, 7) XS_CASE(1, 12) XS_CASE(2, 5) XS_CASE(2, 7) XS_CASE(3, 5)
#undef XS_CASE
  }
#define XB_CASE(UTV, NCHV) \
  if (ut == UTV && nch == NCHV) return launch_expand_bwd_cfg<UTV, NCHV>(A, W, ldw, wrows, X, ldx, ep, dwe, ws, ws_floats, M, inp, hid, (const bf16_t*)mp, ldm, st);
  XB_CASE(1, 5) XB_CASE(1, 7) XB_CASE(1, 12) XB_CASE(2, 5) XB_CASE(2, 7) XB_CASE(3, 5)
#undef XB_CASE
  set_error("expand_bwd: no instance for inp=%d hid=%d", inp, hid);
  return 1;
}

// 1 when atomnas_project_bwd has an instance for this shape: bf16 storage, oup a multiple of 8 and <= 64 (one to four 16-channel
// accumulator tiles of the weight gradient per wave), hid >= 96 and >= 2 * oup
extern "C" int atomnas_project_bwd_supported(int oup, int hid, int dtype) {
  return dtype == DT_BF16 && oup >= 8 && oup <= 64 && oup % 8 == 0 && hid >= 96 && hid >= 2 * oup && hid <= NT_MAX_STAT;
}

// 1 when atomnas_project_bwd serves these layouts: the streaming kernel's conditions (nt_st_kind: switch, 2 GB per 64-channel chunk,
// 4096 tiles per statistics row).  Callers gate the fused form on this query and use the two GEMMs otherwise.
extern "C" int atomnas_project_bwd_dp_supported(long M, int oup, int hid, int ldg, int ldz, long z_ss, int ldgh, long gh_ss, int stat_rows,
                                                int dtype) {
  if (!atomnas_project_bwd_supported(oup, hid, dtype) || M <= 0 || stat_rows <= 0) return 0;
  static float dummy;
  Operand A{&dummy, ldg, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0};
  Epilogue ep{&dummy, ldgh, 0, nullptr, 0, &dummy, ldz, gh_ss, z_ss, nullptr, nullptr, ACT_RELU, nullptr, &dummy, STAT_Z, stat_rows};
  return nt_st_kind(PRO_NONE, A, ep, M, hid, oup) == ST_MASK ? 1 : 0;
}

// Backward of the linear projection nn.Conv2d(hid, oup, 1) (models/mobilenet_base.py:338) in ONE pass over the raw depthwise output:
//   g = dP, the differentiated block-output BatchNorm (atomnas_bnbwd_apply's output; the prologue form that computed it per tile from
//   two streams was removed in round 6)
//   gh[M, hid] = act'(z*zscale + zshift) * (dP * Wp)            -- gradient wrt the raw depthwise-BN output, masked
//   stats rows [sum gh, sum gh*z]                               -- for the depthwise BN's backward
//   dwp[o*si + n*sj] += sum_m dP[m][o] * act(z*zscale + zshift)[m][n]
// wpt: Wp^T packed by atomnas_pack_weights ([hid padded to 64][ldw >= oup rounded up to 32]).
extern "C" int atomnas_project_bwd(const void* g, int ldg, const void* wpt, int ldw, const void* z, int ldz, long z_ss, const float* zscale,
                                   const float* zshift, int act, void* gh, int ldgh, long gh_ss, float* stats, int stat_rows, float* dwp, long si,
                                   long sj, float* ws, long ws_floats, long M, int oup, int hid, int dtype, void* stream) {
  ATOMNAS_REQUIRE(atomnas_project_bwd_supported(oup, hid, dtype), "project_bwd: unsupported shape oup=%d hid=%d dtype=%d", oup, hid, dtype);
  ATOMNAS_REQUIRE(g && wpt && z && zscale && zshift && gh && stats && dwp && ws && M > 0, "project_bwd: bad arguments");
  ATOMNAS_REQUIRE(act >= ACT_RELU && act <= ACT_SWISH && stat_rows > 0, "project_bwd: bad activation / stat_rows");
  ATOMNAS_REQUIRE(ldg >= oup && ldg % 8 == 0, "project_bwd: bad pitch");
  ATOMNAS_REQUIRE((z_ss >= M * 16 || (z_ss == 0 && ldz >= hid && ldz % 8 == 0)) && (gh_ss >= M * 16 || (gh_ss == 0 && ldgh >= hid && ldgh % 8 == 0)),
                  "project_bwd: bad hidden layout");
  ATOMNAS_REQUIRE(ldw >= (oup + 31) / 32 * 32 && ldw % 8 == 0, "project_bwd: packed weight pitch %d too small for oup=%d", ldw, oup);
  Operand A{g, ldg, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0};
  Epilogue ep{gh, ldgh, 0, nullptr, 0, z, ldz, gh_ss, z_ss, zscale, zshift, act, nullptr, stats, STAT_Z, stat_rows};
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* W = (const bf16_t*)wpt;
  const int ut = (oup + 15) / 16;
  ATOMNAS_REQUIRE(nt_st_kind(PRO_NONE, A, ep, M, hid, oup) == ST_MASK, "project_bwd: needs slab-major hidden tensors (or plain ones with hid %% 8 == 0) below 2 GB per 64-channel chunk (ask atomnas_project_bwd_dp_supported)");
  if (ut == 1) return launch_project_bwd_st<1, 1>(A, W, ldw, ep, dwp, si, sj, ws, ws_floats, M, hid, oup, st);
  if (ut == 2) return launch_project_bwd_st<1, 2>(A, W, ldw, ep, dwp, si, sj, ws, ws_floats, M, hid, oup, st);
  if (ut == 3) return launch_project_bwd_st<2, 3>(A, W, ldw, ep, dwp, si, sj, ws, ws_floats, M, hid, oup, st);
  return launch_project_bwd_st<2, 4>(A, W, ldw, ep, dwp, si, sj, ws, ws_floats, M, hid, oup, st);
}

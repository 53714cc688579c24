// Round-4 PROTOTYPE, not part of the product library (built by tools/build_ntswg_experiment.sh, measured by tools/experiments/ntswg_bench.py):
// the wide-input 1x1 GEMM of the LATE stages (projection forward, models/mobilenet_base.py:338,378: N = 80 .. 192 output channels,
// K = 720 .. 3456 hidden channels) on the LDS-DMA queue of k_gemm_nt_sw / k_expand_bwd_s (csrc/pwconv.hip).
//     C[M][N] = act(A * scale + shift) W^T,   statistics rows [sum c, sum c^2]
// Difference to k_gemm_nt_sw: the weights (N x K x 2 bytes, up to 1.3 MB) are not resident.  A compiler-known weight load inside the
// loop would be waited for with a count that ignores the copies and drain the queue, so the 64-channel chunk
// of the weights AND the chunk's scale / shift travel through the same queue as the activations:
//   stage = 8 activation subtiles [32 rows][16 channels] (1 KB each, contiguous in the slab-major tensor)
//         + 2 UT weight tiles of 8 rows x 64 channels (1 KB each; LDS row pitch 128 bytes, the 16-byte pieces of a row XOR-swizzled with
//           the row number -- the copy's LDS destination is lane-linear, so the swizzle is applied to the SOURCE address of each lane)
//         + 1 KB of coefficients (64 scale, 64 shift floats; every wave copies the same bytes: equal copy counts per wave).
// Everything else as in k_gemm_nt_sw: counted vmcnt, one barrier per stage, prologue on the fragment registers, 16 UT output channels.
#include "../common.h"
#include <cstdlib>

namespace atomnas {

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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k0 = 64 * c + 32 * ks + 8 * q;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = cf[ks][e >> 2][e & 3], sh = cf[ks][2 + (e >> 2)][e & 3];
          v[e] = (k0 + e < K) ? (float)hb[ks][e] * sc + sh : 0.f;
        }
        act_apply_v<8>(v, am);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + e >= K) v[e] = 0.f;   // (act(0) is 0 for the three activations; kept explicit)
        bf16x8 af;
#pragma unroll
        for (int e = 0; e < 8; ++e) af[e] = (bf16_t)v[e];
        bf16x8 wf[UT];   // the k-step's weight fragments: all reads in flight, one wait
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          const int wr = 16 * t + j, p = 4 * ks + q;
          const unsigned wa = base + (unsigned)AT * 1024u + (unsigned)wr * 128u + (unsigned)((p ^ (wr & 7)) << 4);
          asm volatile("ds_read_b128 %0, %1" : "=v"(wf[t]) : "v"(wa) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], af, acc[t], 0, 0, 0);
      }
      slot = slot + 1 == DEPTH ? 0 : slot + 1;
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const int c0 = 16 * t + 4 * q;
      if (row < M && c0 < N) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (bf16_t)acc[t][r];
          const float f = (float)o[r];
          ssum[t][r] += f;
          ssq[t][r] += f * f;
        }
        *reinterpret_cast<bf16x4*>(cout + row * ldc + c0) = o;
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
  if (lds > max_lds_bytes()) return 1;
  auto kern = k_gemm_nt_swg<UT, DEPTH, WGPC, NWV>;
  const long rblocks = (M + NWV * 16 - 1) / (NWV * 16);
  long R = (long)num_cus() * resident_per_cu(kern, NWV * 64, lds);
  if (R > rblocks) R = rblocks;
  if (stats && R > stat_rows) R = stat_rows;
  hipLaunchKernelGGL(kern, dim3((unsigned)R), dim3(NWV * 64), lds, st, a, ass, scale, shift, act, W, ldw, c, ldc, stats, stat_rows, M, N, K);
  return check_launch("exp_nt_swg");
}

}  // namespace atomnas

using namespace atomnas;

// a: slab-major bf16 [K / 16][M][16];  W: packed weight [N rounded up to 64][ldw] (atomnas_pack_weights mode 0);  c: plain bf16 [M][ldc];
// N a multiple of 8 and <= 192;  stats (may be NULL): partial rows [stat_rows][2][N]
extern "C" int atomnas_exp_nt_swg(const void* a, long a_ss, const float* scale, const float* shift, int act, const void* w, int ldw, void* c,
                                  int ldc, float* stats, int stat_rows, long M, int N, int K, void* stream) {
  ATOMNAS_REQUIRE(a && a_ss >= M * 16 && scale && shift && w && c && M >= 64 && N >= 8 && N <= 192 && N % 8 == 0 && K >= 64 && K % 4 == 0 &&
                      ldw >= 64 && ldw % 8 == 0 && ldc >= N, "exp_nt_swg: bad arguments");
  const int ut = (N + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  static const int nwv = getenv("ATOMNAS_SWG_WAVES") ? atoi(getenv("ATOMNAS_SWG_WAVES")) : 4;   // 4: 64-row stages, 8: 128-row stages
#define SWG_CASE(UTV, WG)                                                                                                              \
  if (ut == UTV) {                                                                                                                     \
    if (nwv == 8) return launch_swg<UTV, 1, 8>((const bf16_t*)a, a_ss, scale, shift, act, (const bf16_t*)w, ldw, (bf16_t*)c, ldc, stats, stat_rows, M, N, K, st); \
    return launch_swg<UTV, WG, 4>((const bf16_t*)a, a_ss, scale, shift, act, (const bf16_t*)w, ldw, (bf16_t*)c, ldc, stats, stat_rows, M, N, K, st); \
  }
  SWG_CASE(3, 2) SWG_CASE(5, 2) SWG_CASE(6, 2) SWG_CASE(12, 1)
#undef SWG_CASE
  set_error("exp_nt_swg: no instance for N=%d", N);
  return 1;
}

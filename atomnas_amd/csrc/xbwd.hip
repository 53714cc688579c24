// Expand backward WITHOUT the raw expand output E (gfx950, bf16): the inp x inp sized pieces.
//
// The atomic block (models/mobilenet_base.py:316-336, 371-382) expands its narrow input x [M][inp] to the 6x wider E = x We^T and
// normalises it; the backward of that BatchNorm is dE = c1*h + c2*E + c3 (h: the masked gradient of the activated hidden tensor).
// The expand convolution's backward then needs dE in two wide GEMMs (dX = dE We, dWe = dE^T x), which used to read BOTH hidden
// streams h and E.  With E = x We^T the c2 / c3 terms collapse to inp x inp sized corrections:
//       dX  = (c1*h) We + x M + v,          M = We^T diag(c2) We,  v = c3^T We
//       dWe = diag(c1) h^T x + diag(c2) We (X^T X) + c3 (sum x)^T
// so the wide GEMMs read h alone (c1 as their scale prologue: atomnas_expand_bwd with e = NULL, or atomnas_pw_gemm_nt /
// atomnas_pw_gemm_tn with the BNRELU prologue) and this file supplies
//   * atomnas_gram:      G = X^T X and sx = sum x from one pass over the narrow tensor,
//   * atomnas_xb_coeffs: M (packed as a GEMM weight), v, and the last two terms of dWe.
// (The same algebra with the expand recomputed INSIDE the depthwise kernels -- E never in HBM at all -- was built and measured
// in round 4 and lost: csrc/experimental/, DESIGN.md.)  No atomics; all reductions in a fixed order.
#include "common.h"
#include <cstdlib>

namespace atomnas {

// ------------------------------------------------------------------------------------------------- Gram matrix of the block input
// G = X^T X (inp x inp) and sx = sum_m x_m of the narrow block input x [M][ldx] (bf16, inp <= 64, a multiple of 8), as per-WAVE
// partials [G | sx] (added per workgroup in wave order where they fit the LDS) summed in order by k_gram_reduce -- fixed order, no atomics.
// A wave owns 32-row units of x, strided over all waves of the launch.  A unit (UT = inp / 16 rounded up row-major [32][16] subtiles
// of 1 KB) is copied HBM -> LDS by global_load_lds_dwordx4 into the wave's PRIVATE ring, DEPTH - 1 units ahead, and waited for
// with a counted vmcnt: no barrier anywhere.  Both MFMA operands of G's (ti, tj) tile are transposing reads (ds_read_b64_tr_b16) of
// subtiles ti and tj; sx is one more MFMA per subtile against a fragment of ones.  Rows past M are cut off in the A fragments.
// (The first version staged fp32 tiles and multiplied 4 x 4 register blocks on the VALU: 1.1 .. 1.5 TB/s on a tensor of 16 .. 100 MB.)
template <int UT, int DEPTH>
__global__ __launch_bounds__(256) void k_gram_part(const bf16_t* __restrict__ x, int ldx, long M, int inp, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_gr[];   // [4 waves][DEPTH][UT] subtiles of 1 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const long nunits = (M + 31) / 32;
  const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
  const long mine = wid < nunits ? (nunits - wid + nw - 1) / nw : 0;   // units wid, wid + nw, ...
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)s_gr) + (unsigned)wave * (DEPTH * UT * 1024u);
  auto dma = [&](const bf16_t* g, unsigned dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(d) : "memory");
  };
  long iu = 0;
  int islot = 0;
  auto issue_next = [&]() {   // unit iu of this wave (past the end: the last one again, keeps the copy count per unit constant)
    const long u = wid + (iu < mine ? iu : (mine > 0 ? mine - 1 : 0)) * nw;
    long row = (u < nunits ? u : nunits - 1) * 32 + (lane >> 1);
    row = row < M ? row : M - 1;
#pragma unroll
    for (int ct = 0; ct < UT; ++ct) {
      const int ch = 16 * ct + 8 * (lane & 1);
      dma(x + row * ldx + (ch < ldx - 8 ? ch : ldx - 8), lds0 + (unsigned)(islot * UT + ct) * 1024u);
    }
    ATOMNAS_RING_STAGE_END();
    islot = islot + 1 == DEPTH ? 0 : islot + 1;
    ++iu;
  };
  const unsigned tr_lane = (unsigned)(((8 * q + (j >> 2)) * 16 + 4 * (j & 3)) * 2);
  f32x4 gacc[UT][UT], sacc[UT];
#pragma unroll
  for (int a = 0; a < UT; ++a) {
    sacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < UT; ++b) gacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.f;
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) issue_next();
  int slot = 0;
  for (long n = 0; n < mine; ++n) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * UT) : "memory");
    const long rows_left = M - (wid + n * nw) * 32;
    bf16x4 lo[UT], hi[UT];
#pragma unroll
    for (int ct = 0; ct < UT; ++ct) {
      const unsigned a0 = lds0 + (unsigned)(slot * UT + ct) * 1024u + tr_lane;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[ct]) : "v"(a0) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:128" : "=v"(hi[ct]) : "v"(a0) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_next();   // into the slot of the unit before this one (this wave alone reads and writes its ring)
    bf16x8 f[UT], fm[UT];
#pragma unroll
    for (int ct = 0; ct < UT; ++ct) {
      f[ct] = __builtin_shufflevector(lo[ct], hi[ct], 0, 1, 2, 3, 4, 5, 6, 7);
      fm[ct] = f[ct];
      if (rows_left < 32) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (8 * q + e >= rows_left) fm[ct][e] = (bf16_t)0.f;
      }
    }
#pragma unroll
    for (int a = 0; a < UT; ++a) {
      sacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[a], ones, sacc[a], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < UT; ++b) gacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[a], f[b], gacc[a][b], 0, 0, 0);
    }
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus copies must not outlive the workgroup's LDS
  // partial [G | sx]: G[i][jj] at [i * inp + jj], sx[i] at [inp * inp + i]; lane (q, j) of tile (a, b) holds i = 16 a + 4 q + r, jj = 16 b + j.
  // UT <= 3: the four waves' partials are added in wave order through the (now idle) ring and the WORKGROUP writes one partial;
  // UT = 4 (inp > 48: 4 x 16.6 KB do not fit the ring): one partial per wave.
  constexpr bool BLOCK_RED = UT <= 3;
  const long ps = (long)inp * inp + inp;
  float* o = BLOCK_RED ? reinterpret_cast<float*>(s_gr) + wave * ps : ws + wid * ps;
  if (BLOCK_RED) __syncthreads();   // every wave is done with its ring
#pragma unroll
  for (int a = 0; a < UT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * a + 4 * q + r;
      if (i < inp) {
#pragma unroll
        for (int b = 0; b < UT; ++b)
          if (16 * b + j < inp) o[(long)i * inp + 16 * b + j] = gacc[a][b][r];
        if (j == 0) o[(long)inp * inp + i] = sacc[a][r];
      }
    }
  if (BLOCK_RED) {
    __syncthreads();
    const float* sp = reinterpret_cast<const float*>(s_gr);
    float* og = ws + (long)blockIdx.x * ps;
    for (int e = tid; e < ps; e += 256) og[e] = ((sp[e] + sp[ps + e]) + sp[2 * ps + e]) + sp[3 * ps + e];
  }
}
// one wave per output element: lane l adds the partials l, l + 64, ... in order, then a fixed-order butterfly over the lanes
__global__ __launch_bounds__(256) void k_gram_reduce(const float* __restrict__ ws, int parts, int inp, float* __restrict__ gram,
                                                     float* __restrict__ sx) {
  const int n = inp * inp + inp;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= n) return;
  float a = 0.f;
  const int nit = (parts + 63) >> 6;
#pragma unroll 8
  for (int i = 0; i < nit; ++i) {   // unconditional loads (clamped), eight in flight
    const int r = lane + 64 * i;
    const float v = ws[(long)(r < parts ? r : parts - 1) * n + e];
    a += r < parts ? v : 0.f;
  }
  a = wave_sum(a);
  if (lane == 0) {
    if (e < inp * inp) gram[e] = a; else sx[e - inp * inp] = a;
  }
}

// The inp x inp sized corrections of the expand backward without E (see the file header):
//   mp[n][k]      = bf16( sum_c c2_c W[c][n] W[c][k] )          packed for atomnas_pw_gemm_nt ([inp rounded up to 64][ldm], zero padded by the caller)
//   vb[n]         = sum_c c3_c W[c][n]                          (its bias vector)
//   dwe[c*inp+k] += c2_c sum_j W[c][j] G[j][k] + c3_c sx[k]
// Workgroups 0 .. inp-1 own one row n of M (and v[n]): 256 threads = KP columns x S channel subsets, the subsets added in order;
// the remaining workgroups own 256 elements of dwe each.  Fixed-order sums.
__global__ __launch_bounds__(256) void k_xb_coeffs(const float* __restrict__ c2, const float* __restrict__ c3, const bf16_t* __restrict__ wexp,
                                                   int ldwe, const float* __restrict__ gram, int ldg, const float* __restrict__ sx,
                                                   int inp, int C, bf16_t* __restrict__ mp, int ldm,
                                                   float* __restrict__ vb, float* __restrict__ dwe) {
  __shared__ float s_p[256 + 64];
  extern __shared__ float s_g[];   // [inp][inp] Gram matrix + [inp] column sums (the dwe workgroups)
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < inp) {
    const int n = blockIdx.x;
    int KP = 8;
    while (KP < inp) KP *= 2;   // 8 .. 64
    const int S = 256 / KP;
    const int k = tid % KP, sub = tid / KP, kl = k < inp ? k : inp - 1;
    float a = 0.f, va = 0.f;
    // unconditional loads at clamped indices, four channels in flight (behind a branch every load was waited for on its own: 73 us
    // for the 40 x 40 matrix of the 28 x 28 stage)
    const int nit = (C + S - 1) / S;
#pragma unroll 4
    for (int i = 0; i < nit; ++i) {
      const int c = sub + i * S, cl = c < C ? c : C - 1;
      const float wn = (float)wexp[(long)cl * ldwe + n], wk = (float)wexp[(long)cl * ldwe + kl];
      const float q2 = c < C ? c2[cl] : 0.f, q3 = c < C ? c3[cl] : 0.f;
      a += q2 * wn * wk;
      va += q3 * wn;
    }
    s_p[sub * KP + k] = a;
    if (k == 0) s_p[256 + sub] = va;
    __syncthreads();
    if (sub == 0 && k < inp) {
      float t = s_p[k];
      for (int q = 1; q < S; ++q) t += s_p[q * KP + k];
      mp[(long)n * ldm + k] = (bf16_t)t;
    }
    if (tid == 0) {
      float t = s_p[256];
      for (int q = 1; q < S; ++q) t += s_p[256 + q];
      vb[n] = t;
    }
    return;
  }
  for (int i = tid; i < inp * inp + inp; i += 256) s_g[i] = i < inp * inp ? gram[(i / inp) * ldg + i % inp] : sx[i - inp * inp];
  __syncthreads();
  const long e = (long)(blockIdx.x - inp) * 256 + tid;
  if (e >= (long)C * inp) return;
  const int c = (int)(e / inp), k = (int)(e % inp);
  // the channel's weight row in 16-byte pieces (ldwe >= inp rounded up to 32: whole pieces), all in flight, then the products from LDS
  float a = 0.f;
  const bf16_t* wr = wexp + (long)c * ldwe;
  for (int j0 = 0; j0 < inp; j0 += 32) {
    bf16x8 w8[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w8[u] = *reinterpret_cast<const bf16x8*>(wr + j0 + 8 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int jj = j0 + 8 * u + q;
        if (jj < inp) a += (float)w8[u][q] * s_g[jj * inp + k];
      }
  }
  dwe[e] += c2[c] * a + c3[c] * s_g[inp * inp + k];
}

}  // namespace atomnas

using namespace atomnas;

// Gram matrix G = X^T X [inp][inp] and column sums sx [inp] of the block input x [M][ldx] (bf16; inp <= 64, a multiple of 8).
// ws: caller-owned scratch of ws_floats floats for the per-workgroup partials (inp*inp + inp floats each; at least one).
extern "C" int atomnas_gram(const void* x, int ldx, long M, int inp, float* ws, long ws_floats, float* gram, float* sx, int dtype, void* stream) {
  ATOMNAS_REQUIRE(x && ws && gram && sx && M > 0 && inp >= 8 && inp <= 64 && inp % 8 == 0 && ldx >= inp && ldx % 8 == 0 && dtype == DT_BF16,
                  "gram: bad arguments (bf16, inp <= 64 and a multiple of 8)");
  const long ps = (long)inp * inp + inp;
  // one partial per WAVE; two workgroups of four waves per CU, at least four 32-row units per wave
  const long nunits = (M + 31) / 32;
  long blocks = 2L * num_cus();
  if (blocks > (nunits + 15) / 16) blocks = (nunits + 15) / 16;
  if (blocks > ws_floats / (4 * ps)) blocks = ws_floats / (4 * ps);
  ATOMNAS_REQUIRE(blocks >= 1, "gram: workspace too small for four partials (%ld floats)", 4 * ps);
  const long parts = (inp + 15) / 16 <= 3 ? blocks : 4 * blocks;   // k_gram_part: one partial per workgroup (UT <= 3) or per wave
  hipStream_t st = (hipStream_t)stream;
  const int ut = (inp + 15) / 16;
  const bf16_t* xp = (const bf16_t*)x;
#define GRAM_CASE(UTV) \
  if (ut == UTV) hipLaunchKernelGGL((k_gram_part<UTV, 4>), dim3((unsigned)blocks), dim3(256), (size_t)4 * 4 * UTV * 1024, st, xp, ldx, M, inp, ws);
  GRAM_CASE(1) GRAM_CASE(2) GRAM_CASE(3) GRAM_CASE(4)
#undef GRAM_CASE
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((ps + 3) / 4)), dim3(256), 0, st, ws, (int)parts, inp, gram, sx);
  return check_launch("gram");
}

// Corrections of the expand backward without E (see k_xb_coeffs); c2 / c3: BatchNorm-backward coefficients of the C hidden channels.
extern "C" int atomnas_xb_coeffs(const float* c2, const float* c3, const void* wexp, int ldwe, const float* gram, int ldg, const float* sx,
                                 int inp, int C, void* mp, int ldm, float* vb, float* dwe, void* stream) {
  ATOMNAS_REQUIRE(c2 && c3 && wexp && gram && sx && mp && vb && dwe && inp > 0 && inp <= 64 && C > 0 && ldg >= inp && ldwe >= inp && ldm >= inp,
                  "xb_coeffs: bad arguments");
  ATOMNAS_REQUIRE(ldwe >= (inp + 31) / 32 * 32 && ldwe % 8 == 0 && ((size_t)wexp & 15) == 0, "xb_coeffs: the packed weight rows are read in 16-byte pieces (ldwe=%d)", ldwe);
  const long blocks = inp + ((long)C * inp + 255) / 256;
  hipLaunchKernelGGL(k_xb_coeffs, dim3((unsigned)blocks), dim3(256), (size_t)(inp * inp + inp) * sizeof(float), (hipStream_t)stream, c2, c3, (const bf16_t*)wexp, ldwe, gram, ldg, sx,
                     inp, C, (bf16_t*)mp, ldm, vb, dwe);
  return check_launch("xb_coeffs");
}


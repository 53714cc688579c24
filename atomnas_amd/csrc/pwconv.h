// Shared pieces of the pointwise (1x1) GEMM translation units (pwconv.hip: the forward / input-gradient GEMMs and the fused backward
// entry points; pwconv_tn.hip: the weight-gradient GEMMs): MFMA wrappers, operand / epilogue descriptors, prologue loaders, the generic
// epilogue with its statistics.  See pwconv.hip for the design.
#pragma once
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace atomnas {

enum { PRO_NONE = 0, PRO_BNRELU = 1, PRO_BNBWD = 2 };
enum { STAT_NONE = 0, STAT_SQ = 1, STAT_Z = 2 };

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int EPL = 8;  // k-elements per lane per MFMA
  using frag = bf16x8;
  static __device__ __forceinline__ frag pack(const float (&v)[8]) {
    frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16_t)v[i];
    return f;
  }
  static __device__ __forceinline__ frag raw(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int EPL = 1;
  using frag = float;
  static __device__ __forceinline__ frag pack(const float (&v)[1]) { return v[0]; }
  static __device__ __forceinline__ frag raw(const float* p) { return *p; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// Operand description with its prologue.
struct Operand {
  const void* p1; int ld1;   // main stream
  const void* p2; int ld2;   // second stream (PRO_BNBWD: the raw activation x)
  long ss1, ss2;             // slab strides of the two streams (0: plain [M][ld]; see lay_off in common.h)
  const float* c1;           // BNRELU: scale   | BNBWD: c1
  const float* c2;           // BNRELU: shift   | BNBWD: c2
  const float* c3;           //                 | BNBWD: c3
  int relu;                  // BNRELU: apply max(.,0)
};

// Loads EPL consecutive channels starting at channel k of row `row`, applies the prologue, zeroes channels >= K.
template <typename T, int MODE>
__device__ __forceinline__ void load_pro(const Operand& o, long row, bool rowvalid, int k, int K, float (&v)[Mma<T>::EPL]) {
  constexpr int E = Mma<T>::EPL;
  if (!rowvalid || k >= K) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = 0.f;
    return;
  }
  VecIO<T, E>::load(reinterpret_cast<const T*>(o.p1) + lay_off(row, k, o.ld1, o.ss1), v);
  if constexpr (MODE == PRO_BNRELU) {
    float s[E], h[E];
    VecIO<float, E>::load(o.c1 + k, s);
    VecIO<float, E>::load(o.c2 + k, h);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = v[e] * s[e] + h[e];
    act_apply_v<E>(v, act_of(o.relu));
  } else if constexpr (MODE == PRO_BNBWD) {
    float x[E], a1[E], a2[E], a3[E];
    VecIO<T, E>::load(reinterpret_cast<const T*>(o.p2) + lay_off(row, k, o.ld2, o.ss2), x);
    VecIO<float, E>::load(o.c1 + k, a1);
    VecIO<float, E>::load(o.c2 + k, a2);
    VecIO<float, E>::load(o.c3 + k, a3);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = a1[e] * v[e] + a2[e] * x[e] + a3[e];
  }
  if constexpr (E > 1) {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (k + e >= K) v[e] = 0.f;
  }
}

// Same as load_pro for 8 bf16 channels, with the prologue coefficients read from LDS copies (lc1..lc3 point at the lane's first
// channel; entries of channels >= K are zero, which also zeroes those channels).  Per-channel vectors fetched through the
// vector-memory path in an inner loop cost more texture-address cycles than the activation stream itself.
template <int MODE>
__device__ __forceinline__ void load_pro_lds(const Operand& o, long row, bool rowvalid, int k, int K, const float* lc1, const float* lc2,
                                             const float* lc3, float (&v)[8]) {
  if (!rowvalid || k >= K) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    return;
  }
  VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(o.p1) + lay_off(row, k, o.ld1, o.ss1), v);
  if constexpr (MODE == PRO_BNRELU) {
    float s[8], h[8];
    VecIO<float, 8>::load(lc1, s);
    VecIO<float, 8>::load(lc2, h);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * s[e] + h[e];
    act_apply_v<8>(v, act_of(o.relu));
  } else if constexpr (MODE == PRO_BNBWD) {
    float x[8], a1[8], a2[8], a3[8];
    VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(o.p2) + lay_off(row, k, o.ld2, o.ss2), x);
    VecIO<float, 8>::load(lc1, a1);
    VecIO<float, 8>::load(lc2, a2);
    VecIO<float, 8>::load(lc3, a3);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a1[e] * v[e] + a2[e] * x[e] + a3[e];
  }
}

// copies the coefficient vectors of columns col0 .. col0+n-1 into LDS as [3][n] (zeros for columns >= ncols)
template <int MODE>
__device__ __forceinline__ void stage_coeffs(const Operand& o, int col0, int n, int ncols, float* dst, int tid) {
  if constexpr (MODE != PRO_NONE) {
    for (int i = tid; i < 3 * n; i += 256) {
      const int v = i / n, c = col0 + i % n;
      const float* src = (v == 0) ? o.c1 : (v == 1 ? o.c2 : o.c3);
      dst[i] = (c < ncols && src != nullptr && (MODE == PRO_BNBWD || v < 2)) ? src[c] : 0.f;
    }
  }
}

struct Epilogue {
  void* c; int ldc; int out_f32;       // output [M, N] (storage T, or fp32 when out_f32)
  const void* add; int ldadd;          // optional residual stream (storage T, plain layout)
  const void* z; int ldz;              // optional raw activation stream (storage T) for mask / STAT_Z
  long css, zss;                       // slab strides of c and z (0: plain)
  const float* zscale; const float* zshift; int mask;  // mask: c *= [z*zscale+zshift > 0]
  const float* bias;                   // optional per-output-channel bias
  float* stats; int stat_mode;         // [stat_rows][2][N] fp32 partial rows: one row per workgroup row-slot, plain stores (common.h)
  int stat_rows;
};

constexpr int NT_MAX_STAT = 8192;  // widest output for which the epilogue statistics are supported

// Epilogue of one 16-pixel x 64-channel accumulator tile: the lane holds channels nb .. nb+15 of pixel `row`, ordered [t][r].
// bias, residual add, ReLU mask of the producer, rounding to the storage type, store, and the per-channel statistics
// (reduced over the 16 pixels with cross-lane adds, then added by lane j == 0 into the WAVE-PRIVATE LDS row sw[2][swd] at local
// channel nl: plain read-modify-write in program order, no atomics -- the statistics are bit-reproducible).
template <typename T>
__device__ __forceinline__ void nt_epilogue_core(const Epilogue& ep, const f32x4 (&acc)[4], long row, bool rowvalid, int nb, int N,
                                                 float (&c)[16], float (&zv)[16]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[4 * t + r] = acc[t][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) zv[i] = 0.f;

  if (rowvalid) {
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
      const int n8 = nb + 8 * h8;
      if (n8 >= N) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[8 * h8 + i] = 0.f;
        continue;
      }
      float tmp[8];
      if (ep.bias) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[8 * h8 + i] += (n8 + i < N) ? ep.bias[n8 + i] : 0.f;
      }
      if (ep.add) {
        VecIO<T, 8>::load(reinterpret_cast<const T*>(ep.add) + row * ep.ldadd + n8, tmp);
#pragma unroll
        for (int i = 0; i < 8; ++i) c[8 * h8 + i] += tmp[i];
      }
      if (ep.z) {
        VecIO<T, 8>::load(reinterpret_cast<const T*>(ep.z) + lay_off(row, n8, ep.ldz, ep.zss), tmp);
#pragma unroll
        for (int i = 0; i < 8; ++i) zv[8 * h8 + i] = tmp[i];
        if (ep.mask) {
          float a8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int n = n8 + i;
            a8[i] = (n < N) ? tmp[i] * ep.zscale[n] + ep.zshift[n] : 0.f;
          }
          act_bwd_v<8>(&c[8 * h8], a8, act_of(ep.mask));
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (n8 + i >= N) c[8 * h8 + i] = 0.f;
      if (ep.out_f32) {
        float* cp = reinterpret_cast<float*>(ep.c) + lay_off(row, n8, ep.ldc, ep.css);
        float o8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] = c[8 * h8 + i];
        VecIO<float, 8>::store(cp, o8);
      } else if constexpr (sizeof(T) == 2) {
        // one packed conversion per pair; the statistics see the stored values through the bits that are stored (common.h: bf16_pair_f32)
        bf16x8 ob;
#pragma unroll
        for (int i = 0; i < 8; ++i) ob[i] = (bf16_t)c[8 * h8 + i];
        typedef __attribute__((ext_vector_type(4))) unsigned bits4;
        bits4 bits = __builtin_bit_cast(bits4, ob);
        asm volatile("" : "+v"(bits));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 o = bf16_pair_f32(bits[i]);
          c[8 * h8 + 2 * i] = o[0];
          c[8 * h8 + 2 * i + 1] = o[1];
        }
        *reinterpret_cast<bits4*>(reinterpret_cast<T*>(ep.c) + lay_off(row, n8, ep.ldc, ep.css)) = bits;
      } else {
        float o8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o8[i] = to_f32(from_f32<T>(c[8 * h8 + i]));
          c[8 * h8 + i] = o8[i];  // statistics see the stored value
        }
        VecIO<T, 8>::store(reinterpret_cast<T*>(ep.c) + lay_off(row, n8, ep.ldc, ep.css), o8);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
  }
}

template <typename T>
__device__ __forceinline__ void nt_epilogue(const Epilogue& ep, const f32x4 (&acc)[4], long row, bool rowvalid, int nb, int N,
                                            bool do_stats, float* sw, int swd, int nl, int j) {
  float c[16], zv[16];
  nt_epilogue_core<T>(ep, acc, row, rowvalid, nb, N, c, zv);
  if (do_stats) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s1 = c[i];
      float s2 = (ep.stat_mode == STAT_SQ) ? c[i] * c[i] : c[i] * zv[i];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
      }
      if (j == 0 && nb + i < N) {
        sw[nl + i] += s1;
        sw[swd + nl + i] += s2;
      }
    }
  }
}

// Statistics flush shared by k_gemm_nt and k_gemm_nt_ws: the four wave-private rows s_stat[4][2][swd] are added in wave order and
// stored as row `rs` of the partial-row buffer for channels nc0 .. nc0+swd-1; rows rs+R, rs+2R, ... are zero-filled.
__device__ __forceinline__ void nt_flush_stats(const Epilogue& ep, const float* s_stat, int swd, int nc0, int N, int rs, int R, int tid) {
  __syncthreads();
  for (int i = tid; i < 2 * swd; i += 256) {
    const int pl = i / swd, c = nc0 + i % swd;
    if (c < N) {
      const float v = ((s_stat[i] + s_stat[2 * swd + i]) + s_stat[4 * swd + i]) + s_stat[6 * swd + i];
      const long elem = (long)pl * N + c;
      ep.stats[(long)rs * 2 * N + elem] = v;
      stat_zero_tail(ep.stats, 2L * N, rs + R, R, ep.stat_rows, elem);
    }
  }
}

struct TrFrag { bf16x4 lo, hi; };   // the two halves (k = 8 q .. + 3, + 4 .. + 7) of a transposing-read MFMA fragment

static inline int check_operand(const char* who, const Operand& o, int mode, int C) {
  ATOMNAS_REQUIRE(o.p1 != nullptr && (o.ss1 > 0 || (o.ld1 >= C && o.ld1 % 8 == 0)), "%s: bad main stream (ld=%d, C=%d)", who, o.ld1, C);
  if (mode == PRO_BNRELU) ATOMNAS_REQUIRE(o.c1 && o.c2, "%s: BNRELU prologue needs scale and shift", who);
  if (mode == PRO_BNBWD)
    ATOMNAS_REQUIRE(o.p2 && (o.ss2 > 0 || (o.ld2 >= C && o.ld2 % 8 == 0)) && o.c1 && o.c2 && o.c3, "%s: BNBWD prologue needs x, c1, c2, c3", who);
  return 0;
}


}  // namespace atomnas

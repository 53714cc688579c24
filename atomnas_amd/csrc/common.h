// Shared device helpers for the AtomNAS gfx950 kernels (CDNA4, wave64).
// Activations are NHWC with an explicit channel pitch `ld` (elements, multiple of 8 so that
// every pixel row starts 16-byte aligned for bf16); storage type T is float or bf16_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atomnas {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { DT_F32 = 0, DT_BF16 = 1 };

// Per-channel statistics leave a kernel as PARTIAL ROWS: a buffer [stat_rows][2][pitch] in which every workgroup (or wave)
// that reduces a channel range owns one row and writes it with plain stores -- no atomics, so the result does not depend on
// arrival order and a training step is bit-reproducible (and same-address atomics from thousands of workgroups serialise at
// ~350 ns each on MI355X).  A producer that uses R < stat_rows rows zero-fills rows R..stat_rows-1 of its channel range
// (stat_zero_tail), so the buffer needs no initialisation and the BatchNorm finalize kernels sum all stat_rows rows in a
// fixed order.
__device__ __forceinline__ void stat_zero_tail(float* __restrict__ stats, long row_stride, int first_row, int row_step, int stat_rows,
                                               long elem) {
  for (int r = first_row; r < stat_rows; r += row_step) stats[(long)r * row_stride + elem] = 0.f;
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE (v_cvt_pk_bf16_f32)

// Load / store V consecutive channels of storage type T as fp32 registers.
template <typename T, int V> struct VecIO;

template <int V> struct VecIO<float, V> {
  static __device__ __forceinline__ void load(const float* p, float (&o)[V]) {
    if constexpr (V % 4 == 0) {
#pragma unroll
      for (int i = 0; i < V / 4; ++i) {
        f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * i);
        o[4 * i] = t[0]; o[4 * i + 1] = t[1]; o[4 * i + 2] = t[2]; o[4 * i + 3] = t[3];
      }
    } else if constexpr (V % 2 == 0) {
#pragma unroll
      for (int i = 0; i < V / 2; ++i) {
        f32x2 t = *reinterpret_cast<const f32x2*>(p + 2 * i);
        o[2 * i] = t[0]; o[2 * i + 1] = t[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = p[i];
    }
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[V]) {
    if constexpr (V % 4 == 0) {
#pragma unroll
      for (int i = 0; i < V / 4; ++i) {
        f32x4 t = {o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
        *reinterpret_cast<f32x4*>(p + 4 * i) = t;
      }
    } else if constexpr (V % 2 == 0) {
#pragma unroll
      for (int i = 0; i < V / 2; ++i) {
        f32x2 t = {o[2 * i], o[2 * i + 1]};
        *reinterpret_cast<f32x2*>(p + 2 * i) = t;
      }
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) p[i] = o[i];
    }
  }
};

template <int V> struct VecIO<bf16_t, V> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[V]) {
    if constexpr (V == 8) {
      bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (float)t[i];
    } else if constexpr (V == 4) {
      bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (float)t[i];
    } else if constexpr (V == 2) {
      bf16x2 t = *reinterpret_cast<const bf16x2*>(p);
      o[0] = (float)t[0]; o[1] = (float)t[1];
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = (float)p[i];
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&o)[V]) {
    if constexpr (V == 8) {
      bf16x8 t;
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = (bf16_t)o[i];
      *reinterpret_cast<bf16x8*>(p) = t;
    } else if constexpr (V == 4) {
      bf16x4 t;
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = (bf16_t)o[i];
      *reinterpret_cast<bf16x4*>(p) = t;
    } else if constexpr (V == 2) {
      bf16x2 t; t[0] = (bf16_t)o[0]; t[1] = (bf16_t)o[1];
      *reinterpret_cast<bf16x2*>(p) = t;
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) p[i] = (bf16_t)o[i];
    }
  }
};

// Activation modes carried by the integer `relu` / `mask` arguments of the C ABI: 0 none, 1 ReLU, 2 ReLU6, 3 Swish
// (models/mobilenet_base.py:407-415 `get_active_fn`, :72-80 `Swish`).
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_SWISH = 3 };
// Resolved once per kernel into two uniform bounds: none (-inf, +inf), ReLU (0, +inf), ReLU6 (0, 6).  The forward is ONE v_med3_f32 per
// element for the three of them (round 6: the earlier `fminf(relu ? fmaxf(a, 0) : a, hi)` compiled to five instructions per element --
// two canonicalising v_max, v_max with 0, a select on the flag, v_min -- 40 of the ~90 prologue instructions per k-step of the 1x1 GEMMs,
// which are bound by instruction issue where they run at one wave per SIMD: profiles/r06_late_stage_gemm_experiments.txt); the backward is
// two compares.  Same values for every non-NaN input; a NaN now propagates instead of becoming 0.
// Swish (x * sigmoid(x)) sits behind a wave-uniform branch: the ReLU paths do not execute the exponential.
struct Act { float lo; float hi; int swish; };
__device__ __forceinline__ Act act_of(int mode) {
  return Act{mode == ACT_RELU || mode == ACT_RELU6 ? 0.f : -__builtin_inff(), mode == ACT_RELU6 ? 6.f : __builtin_inff(), mode == ACT_SWISH};
}
__device__ __forceinline__ float swish_f(float a) { return a / (1.f + __expf(-a)); }
// d/da [a * sigmoid(a)] = s * (1 + a * (1 - s))
__device__ __forceinline__ float swish_grad(float a) {
  const float s = 1.f / (1.f + __expf(-a));
  return s * (1.f + a * (1.f - s));
}
__device__ __forceinline__ float act_apply(float a, Act m) {
  if (__builtin_expect(m.swish, 0)) return swish_f(a);
  return __builtin_amdgcn_fmed3f(a, m.lo, m.hi);
}
// vector form: ONE wave-uniform branch per V values (per-element calls leave a branch per element in the unrolled hot loops)
template <int V> __device__ __forceinline__ void act_apply_v(float (&a)[V], Act m) {
  if (__builtin_expect(m.swish, 0)) {
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = swish_f(a[e]);
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = __builtin_amdgcn_fmed3f(a[e], m.lo, m.hi);
  }
}
// act_pass = the derivative is non-zero at pre-activation a (ReLU / ReLU6 / none)
__device__ __forceinline__ bool act_pass(float a, Act m) { return a > m.lo && a < m.hi; }
// gradient c of the activated value back through the activation at pre-activation a
__device__ __forceinline__ float act_bwd(float c, float a, Act m) {
  if (__builtin_expect(m.swish, 0)) return c * swish_grad(a);
  return act_pass(a, m) ? c : 0.f;
}
template <int V> __device__ __forceinline__ void act_bwd_v(float* c, const float (&a)[V], Act m) {
  if (__builtin_expect(m.swish, 0)) {
#pragma unroll
    for (int e = 0; e < V; ++e) c[e] *= swish_grad(a[e]);
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) c[e] = act_pass(a[e], m) ? c[e] : 0.f;
  }
}

// fp32 values of a PACKED pair of bf16 (bits of two stored outputs): one shift, one mask.  The epilogues that accumulate statistics of
// the STORED values use it on the registers they store: written as `(float)(bf16_t)x` per element the compiler converts every value twice
// (one v_cvt_pk_bf16_f32 per pair for the store, one per ELEMENT for the statistics -- 6 conversions per 4 outputs, 17 % of the expand
// forward's loop; round 6, profiles/r06_late_stage_gemm_experiments.txt).  Callers pin the packed registers with an empty asm BEFORE the
// store and the statistics read them, so that instruction selection cannot split the conversion again.
__device__ __forceinline__ f32x2 bf16_pair_f32(unsigned bits) {
  return f32x2{__builtin_bit_cast(float, bits << 16), __builtin_bit_cast(float, bits & 0xffff0000u)};
}

// Activation layouts (include/atomnas_hip.h).  plain: [M][ld], element (row, c) at row*ld + c.  slab-major (ss > 0): the channel
// dimension is cut into slabs of 16 channels and every slab is a contiguous [M][16] matrix, slab stride ss elements:
// (c/16)*ss + row*16 + c%16.  A workgroup that owns a channel range then streams CONTIGUOUS memory (32-byte pixels back to back)
// instead of 32-byte fragments at the pixel pitch -- measured with tools/probe/membw.hip: 2.6-2.8 TB/s (fragments of a 432-channel
// tensor) vs 5.3 TB/s (contiguous) for a 3-reads-1-write pass.  8-channel groups (16-byte accesses) never straddle a slab.
__device__ __forceinline__ long lay_off(long row, int c, int ld, long ss) {
  return ss ? (long)(c >> 4) * ss + row * 16 + (c & 15) : row * ld + c;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// status plumbing shared by all translation units of the C-ABI library
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Launch sizing.  The hot kernels are persistent: their grids are exactly the number of workgroups that are resident at
// once (measured: a partial second round of workgroups cost the depthwise kernels up to 1.8x).  Host-side queries only,
// cached per (kernel, dynamic LDS size): nothing is enqueued, so launches stay capturable into a hipGraph.
int num_cus();
size_t max_lds_bytes();   // LDS bytes per workgroup on the current device (160 KiB on gfx950), queried
int resident_per_cu_raw(const void* kern, int threads, size_t lds);
template <typename KernelT>
static inline int resident_per_cu(KernelT kern, int threads, size_t lds) {
  return resident_per_cu_raw((const void*)kern, threads, lds);
}

// Fixed-order reduction of per-workgroup partial results (weight gradients):
//   out[(e / inner) * s_outer + (e % inner) * s_inner] += sum_{r < parts} part[r * part_stride + e]   for e < n
// The order of the additions depends only on (parts, n): bit-reproducible, unlike atomic accumulation.
int reduce_parts(const float* part, long part_stride, int parts, long n, float* out, int inner, long s_outer, long s_inner,
                 hipStream_t st);

// dwconv_cw.hip: the stride-1 / slab-major instances of the depthwise entry points.  Return -1 when the case is not theirs
// (the caller goes on with the tile kernels of dwconv.hip), otherwise the launch status.
int dwconv_cw_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                  float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k, int dtype, hipStream_t st);
int dwconv_cw_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                  const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                  float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k, int stride,
                  int dtype, hipStream_t st);

// dwconv_mm.hip: the bf16 instances of the depthwise forward with the tap arithmetic on the matrix cores (same contract: -1 = not mine)
int dwconv_mm_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                  float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k, int dtype, hipStream_t st);
int dwconv_mm_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3,
                  const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h, long hss,
                  float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k, int stride,
                  int dtype, hipStream_t st);
int dwconv_mm_supported(int N, int H, int W, int C, int k, int dir);
// dwconv_mm2.hip: the stride-2 forward on the matrix cores
int dwconv_mm2_fwd(const void* x, long xss, const float* sc, const float* sh, int relu, const float* w, int ldw, void* y, long yss,
                   float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k, int dtype, hipStream_t st);
int dwconv_mm2_supported(int N, int H, int W, int C, int k);

// End of one stage of an LDS-DMA ring (all of the stage's global_load_lds copies of this wave have been issued).  An assembler
// comment, no code: tools/check_asm_waits.py models the ring as a queue of stages and checks every counted `s_waitcnt vmcnt(N)`
// against the number of copies issued AFTER the end of the stage that wait is for.
#define ATOMNAS_RING_STAGE_END() asm volatile("; atomnas_ring_stage_end" ::: "memory")

#define ATOMNAS_REQUIRE(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      ::atomnas::set_error(__VA_ARGS__);      \
      return 1;                               \
    }                                         \
  } while (0)

}  // namespace atomnas

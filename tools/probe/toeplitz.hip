// Micro-benchmark (gfx950): the tap arithmetic of a k x k depthwise convolution as MFMAs against a Toeplitz operand, next to the packed-FMA
// tap rows the channel-pair kernels use (csrc/dwconv_cw.hip k_dwf_cw).  VERDICT r3 item 2a.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/toeplitz.hip -o gpurun_out/toeplitz && gpurun_out/toeplitz
//
// Formulation (one channel, one tap row ky): out[row][x0 + i] += sum_k T[i][k] * in[row + ky][x0 + k],  T[i][k] = w[ky][k - i] for
// 0 <= k - i < K.  As mfma_f32_16x16x32_bf16:  A = T (16 outputs x 32 window columns, built once per channel and tap row from the
// taps), B = the input window (32 columns x 16 ROWS of the image: lane j reads 8 consecutive bf16 of row j: one aligned ds_read_b128),
// D = 16 output columns x 16 rows.  K MFMAs (one per tap row) per 256 outputs and channel: 112 matrix-pipe cycles for k = 7, where
// the packed-FMA form issues 343 v_pk_fma_f32 per 448 pixel pairs.  The price: the operand planes must be bf16 (the activated input
// is rounded once more), the taps too.
// The program checks the MFMA form against a direct convolution on the host, then times both forms (cycles per output pixel and
// channel PAIR, 1 / 2 / 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int ROWS = 16, COLS = 64;          // output tile of one wave: 16 rows x 64 columns (4 MFMA column tiles)
template <int K> struct Geo {
  static constexpr int LH = ROWS + K - 1;    // window rows
  static constexpr int PITCH = 72;           // bf16 elements per window row (64 + K - 1 <= 70): 36 dwords, so the 16 rows of a b128 read group hit 16 different 16-byte slots
};

// ---- MFMA form.  planes: [2 channels][LH][PITCH] bf16 in LDS per wave; taps: [2][K][K] floats in global.
template <int K>
__global__ __launch_bounds__(256) void k_mfma(const bf16_t* __restrict__ in, const float* __restrict__ w, float* __restrict__ out,
                                              unsigned long long* cyc, int iters) {
  using G = Geo<K>;
  extern __shared__ __attribute__((aligned(16))) bf16_t s_in[];   // [waves][2][LH][PITCH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  bf16_t* pl = s_in + wave * 2 * G::LH * G::PITCH;
  for (int i = lane; i < 2 * G::LH * G::PITCH; i += 64) pl[i] = in[i];
  __syncthreads();
  // Toeplitz fragments: lane (i = j, k = 8 q + e): w[ky][k - i]
  bf16x8 tf[2][K];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = 8 * q + e - j;
        tf[c][ky][e] = (bf16_t)((d >= 0 && d < K) ? w[(c * K + ky) * K + d] : 0.f);
      }
  f32x4 acc[2][4];
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(pl + (c * G::LH + j + ky) * G::PITCH + 16 * t + 8 * q);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[c][ky], b, a, 0, 0, 0);
        }
        acc[c][t] = a;
        asm volatile("" : "+v"(acc[c][t]));
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  if (blockIdx.x == 0 && wave == 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(c * ROWS + j) * COLS + 16 * t + 4 * q + r] = acc[c][t][r];   // [c][row j][x]
  }
}

// ---- packed-FMA form of k_dwf_cw: lanes = 7-pixel strips (8 rows x 8 strips of a 56-wide tile), fp32 pair planes, taps as scalars
template <int K>
__global__ __launch_bounds__(256) void k_valu(const float* __restrict__ w, float* __restrict__ out, unsigned long long* cyc, int iters) {
  constexpr int LH = 8 + K - 1, LWP = 72;
  extern __shared__ __attribute__((aligned(16))) f32x2 s_p[];   // [waves][LH][LWP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x2* pl = s_p + wave * LH * LWP;
  for (int i = lane; i < LH * LWP; i += 64) pl[i] = f32x2{(float)(i % 13) * 0.1f, (float)(i % 7) * 0.2f};
  __syncthreads();
  const int r = lane >> 3, sj = lane & 7;
  const unsigned addr0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(pl + r * LWP + 7 * sj));
  f32x2 acc[7];
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 7; ++t) acc[t] = f32x2{0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      f32x2 inb[7 + K - 1], wb[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) asm volatile("s_load_dwordx2 %0, %1, %2" : "=&s"(wb[kx]) : "s"(w), "s"((unsigned)((ky * K + kx) * 8)));
#pragma unroll
      for (int i = 0; i < 7 + K - 1; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(inb[i]) : "v"(addr0 + (unsigned)(ky * LWP * 8)), "i"(i * 8));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] += inb[t + kx] * wb[kx];
#pragma unroll
      for (int t = 0; t < 7; ++t) asm volatile("" : "+v"(acc[t]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  if (blockIdx.x == 0 && wave == 0) for (int t = 0; t < 7; ++t) { out[(lane * 7 + t) * 2] = acc[t][0]; out[(lane * 7 + t) * 2 + 1] = acc[t][1]; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int K>
void run() {
  using G = Geo<K>;
  const int nin = 2 * G::LH * G::PITCH;
  std::vector<bf16_t> hin(nin);
  std::vector<float> hinf(nin), hw(2 * K * K);
  srand(K);
  for (int i = 0; i < nin; ++i) { float v = (rand() % 2001 - 1000) / 500.f; hin[i] = (bf16_t)v; hinf[i] = (float)hin[i]; }
  for (auto& v : hw) { float t = (rand() % 2001 - 1000) / 1000.f; v = (float)(bf16_t)t; }
  bf16_t* din; float *dw, *dout; unsigned long long* dc;
  CK(hipMalloc(&din, nin * 2)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dout, 2 * ROWS * COLS * 4 + 4096)); CK(hipMalloc(&dc, 4096 * 8));
  CK(hipMemcpy(din, hin.data(), nin * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  // correctness of the Toeplitz form
  const size_t ldsm = (size_t)4 * nin * 2;
  hipLaunchKernelGGL(k_mfma<K>, dim3(1), dim3(256), ldsm, 0, din, dw, dout, dc, 1);
  CK(hipDeviceSynchronize());
  std::vector<float> ho(2 * ROWS * COLS);
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < ROWS; ++r)
      for (int x = 0; x < COLS; ++x) {
        double ref = 0;
        for (int ky = 0; ky < K; ++ky)
          for (int kx = 0; kx < K; ++kx) ref += (double)hw[(c * K + ky) * K + kx] * hinf[(c * G::LH + r + ky) * G::PITCH + x + kx];
        maxerr = fmax(maxerr, fabs(ref - ho[(c * ROWS + r) * COLS + x])); maxref = fmax(maxref, fabs(ref));
      }
  printf("k=%d: Toeplitz MFMA form vs direct convolution: max |err| %.3e (max |ref| %.2f)\n", K, maxerr, maxref);
  const int iters = 200;
  for (int wps = 1; wps <= 4; wps *= 2) {
    // wps waves per SIMD: wps workgroups of 4 waves per CU, 256 CUs
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k_mfma<K>, dim3(blocks), dim3(256), ldsm, 0, din, dw, dout, dc, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hc(blocks * 4);
    CK(hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost));
    double cm = 0; for (auto v : hc) cm += (double)v; cm /= hc.size();
    const size_t ldsv = (size_t)4 * (8 + K - 1) * 72 * 8;
    hipLaunchKernelGGL(k_valu<K>, dim3(blocks), dim3(256), ldsv, 0, dw, dout, dc, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost));
    double cv = 0; for (auto v : hc) cv += (double)v; cv /= hc.size();
    // counter ticks per output pixel PAIR and wave, times waves per SIMD = SIMD time per pixel pair
    const double pm = cm / iters / (ROWS * COLS) / wps, pv = cv / iters / 448.0 / wps;
    printf("  %d wave(s)/SIMD: MFMA form %.4f ticks per pixel pair and SIMD, packed-FMA form %.4f  -> ratio %.2f\n", wps, pm, pv, pv / pm);
  }
  CK(hipFree(din)); CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(dc));
}

int main() {
  run<3>(); run<5>(); run<7>();
  return 0;
}

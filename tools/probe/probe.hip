// Toolchain probe: C-ABI kernel launched on a torch stream via ctypes.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k_axpy(float* y, const float* x, float a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

// One wave computes D[16x16] = A[16x32] * B[32x16] with mfma 16x16x32 bf16.
// A row-major [16][32], Bt row-major [16][32] (i.e. B^T), D row-major [16][16] fp32.
__global__ void k_mfma(const __bf16* A, const __bf16* Bt, float* D) {
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l & 15) * 32 + 8 * (l >> 4));
  bf16x8 b = *(const bf16x8*)(Bt + (l & 15) * 32 + 8 * (l >> 4));
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

extern "C" int probe_axpy(float* y, const float* x, float a, int n, void* stream) {
  hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  return (int)hipGetLastError();
}
extern "C" int probe_mfma(const void* A, const void* Bt, float* D, void* stream) {
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, (const __bf16*)A, (const __bf16*)Bt, D);
  return (int)hipGetLastError();
}
extern "C" int probe_runtime_version() { int v = 0; hipRuntimeGetVersion(&v); return v; }

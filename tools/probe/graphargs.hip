// Experiment: does hipGraph replay issue a device copy per kernel node whose by-value argument block is large?
// (The bs-256 step shows ~280 __amd_rocclr_copyBuffer per replay = the number of launches with DwGeom / Operand / Epilogue structs.)
//   rocprofv3 --kernel-trace --stats -- tools/variants/graphargs <bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int N> struct Blob { int v[N]; };
template <int N> __global__ void k_args(float* out, Blob<N> b) { if (threadIdx.x == 0) out[blockIdx.x] = (float)b.v[N - 1]; }
template <int N> void run(float* out, hipStream_t st) {
  Blob<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_args<N>, dim3(4), dim3(64), 0, st, out, b);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  printf("args %d bytes: 100 nodes x 10 replays done\n", (int)sizeof(Blob<N>) + 8);
}
int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 0;
  float* out; hipMalloc(&out, 4096);
  hipStream_t st; hipStreamCreate(&st);
  if (which == 0) run<4>(out, st);        // 24 bytes
  else if (which == 1) run<24>(out, st);  // 104 bytes
  else if (which == 2) run<48>(out, st);  // 200 bytes
  else run<120>(out, st);                 // 488 bytes
  return 0;
}

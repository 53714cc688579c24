// Experiment (not product): streaming rate of the depthwise kernels' ACCESS ORDER, without their arithmetic.
// y[n][h][w][c] = x[n][h][w][c] over NHWC bf16; a workgroup owns a CB-channel slab and walks spatial tiles TH x TW (reading
// a halo of HALO pixels around each tile like the depthwise kernels do, writing the tile itself), 16-byte pieces per lane,
// same XCD-aware slab placement and contiguous tile ranges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void k_tilecopy(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int N, int H, int W,
                                                  int C, int CB, int TH, int TW, int HALO, int nslabs, int nworkers, int nstreams,
                                                  size_t stream_stride) {
  const int CG = CB / 8;
  const int b_xcd = blockIdx.x & 7, b_local = blockIdx.x >> 3;
  const int slab = b_local % nslabs, worker = (b_local / nslabs) * 8 + b_xcd;
  if (worker >= nworkers) return;
  const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
  const int ntiles = N * tiles_y * tiles_x;
  const int t_beg = (int)((long)worker * ntiles / nworkers), t_end = (int)((long)(worker + 1) * ntiles / nworkers);
  const int cg = threadIdx.x % CG;
  const long coff = (long)slab * CB + cg * 8;
  const int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  u32x4 acc = {0, 0, 0, 0};
  for (int t = t_beg; t < t_end; ++t) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
    // halo'd reads (nstreams input tensors)
    for (int s = 0; s < nstreams; ++s) {
      const unsigned short* xs = x + s * stream_stride;
      for (int pix = threadIdx.x / CG; pix < LH * LW; pix += 256 / CG) {
        const int hi = ty * TH - HALO + pix / LW, wi = tx * TW - HALO + pix % LW;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) acc += *reinterpret_cast<const u32x4*>(xs + (((long)n * H + hi) * W + wi) * C + coff);
      }
    }
    // tile write
    for (int pix = threadIdx.x / CG; pix < TH * TW; pix += 256 / CG) {
      const int hi = ty * TH + pix / TW, wi = tx * TW + pix % TW;
      if (hi < H && wi < W) *reinterpret_cast<u32x4*>(y + (((long)n * H + hi) * W + wi) * C + coff) = acc;
    }
  }
}

int main() {
  const int N = 256, H = 56, W = 56, C = 144, CB = 16;
  const size_t elems = (size_t)N * H * W * C;
  unsigned short *x, *y;
  const int NBUF = 2;
  hipMalloc(&x, (elems + 200000) * 2 * 3 * NBUF + 4096); hipMalloc(&y, (elems + 600000) * 2 * NBUF);
  hipMemset(x, 1, (elems + 200000) * 2 * 3 * NBUF); hipMemset(y, 0, (elems + 600000) * 2 * NBUF);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { int th, tw, halo, streams, percu; size_t pad; };   // pad: extra elements between the input streams (HBM channel skew)
  const Cfg cfgs[] = {{14, 14, 1, 3, 3, 0}, {14, 14, 1, 3, 3, 2048}, {14, 14, 1, 3, 3, 6144}, {14, 14, 1, 3, 3, 34816}, {14, 14, 1, 3, 3, 133120},
                      {14, 14, 3, 3, 2, 0}, {14, 14, 3, 3, 2, 6144}, {14, 14, 3, 3, 2, 133120}, {14, 14, 1, 1, 4, 0}, {14, 14, 1, 1, 4, 6144}};
  for (const Cfg& c : cfgs) {
    const int nslabs = C / CB;
    const int nworkers = 256 * c.percu / nslabs;
    const int grid = (nworkers + 7) / 8 * 8 * nslabs;
    auto launch = [&](int it) {
      hipLaunchKernelGGL(k_tilecopy, dim3(grid), dim3(256), 0, 0, x + (size_t)(it % NBUF) * (elems + 200000) * 3, y + (size_t)(it % NBUF) * elems + c.pad * 3, N, H, W, C, CB,
                         c.th, c.tw, c.halo, nslabs, nworkers, c.streams, elems + c.pad);
    };
    for (int i = 0; i < 2; ++i) launch(i);
    hipDeviceSynchronize();
    const int IT = 8;
    hipEventRecord(e0);
    for (int i = 0; i < IT; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= IT;
    const double moved = (double)elems * 2 * (c.streams + 1);
    printf("tile %2dx%2d halo %d streams %d percu %d pad %6zu : %.3f ms  %.0f GB/s (tensor bytes)\n", c.th, c.tw, c.halo, c.streams, c.percu, c.pad, ms,
           moved / ms / 1e6);
  }
  return 0;
}

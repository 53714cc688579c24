// Input pipeline on the GPU: decoded uint8 HWC images -> crop -> bilinear / bicubic resize -> horizontal flip -> ToTensor ->
// Normalize, straight into the batch tensor the stem reads.  What the reference does per sample on CPU workers with PIL / torchvision
// (utils/dataflow.py:92-170 `data_transforms` 'imagenet1k_mnas_bilinear' / 'imagenet1k_mnas_bicubic' -- the latter is the default of
// apps/mobilenet/default_mnas_scheduler.yml --: RandomResizedCropPadding / CenterCropPadding + Resize,
// RandomHorizontalFlip, ToTensor, Normalize; utils/transforms.py:79-177) -- the random crop PARAMETERS stay host logic
// (atomnas_amd/utils/transforms.py restates them), the pixel work is this kernel.  JPEG decoding and LMDB are out of scope (no decoder
// in the image).
//
// The resize is PIL's (Image.resize(size, Image.BILINEAR | Image.BICUBIC) on the cropped image, which is what torchvision's
// F.resized_crop / Resize call): a separable triangle filter (support 1) or Keys cubic with a = -0.5 (support 2: negative lobes, hence the
// clamps of both passes) whose support grows with the down-scaling factor (antialiasing), coefficients normalised and
// quantised to 22 fractional bits, horizontal pass first, each pass rounded to uint8 (libImaging/Resample.c: precompute_coeffs,
// normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc).  Restated here operation by operation in double / int32
// arithmetic with FP contraction off, so the result is BIT-IDENTICAL to PIL's (tests/test_input_pipeline_gpu.py against fixtures
// generated with PIL in this container, tools/make_golden_input.py).  ToTensor / Normalize: (u8 / 255 - mean) / std in fp32, the
// order torchvision applies.
//
// One thread per output pixel: it builds its own <= KMAX horizontal and vertical coefficients (a few dozen double operations) and walks
// its (rows x columns) support window; at the usual scales (1.2 .. 2.5) that is 5 x 5 taps x 3 channels.  Reads are uint8 from the
// packed image pool (L2-resident window per output row), writes are coalesced along x.  A batch of 256 x 224 x 224 takes tens of
// microseconds: the H2D copy of the uint8 pixels is what the prefetcher overlaps with the step.
#include "common.h"

namespace atomnas {

struct ImgDesc {
  long off;          // byte offset of the image in the pool (HWC, 3 channels, row pitch = 3 W)
  int H, W;          // decoded size
  int bi, bj, bh, bw;   // crop box: top, left, height, width (inside the image)
  int flip;          // mirror the result horizontally
  int pad_;
};
static_assert(sizeof(ImgDesc) == 40, "atomnas_img_desc layout");

constexpr int PP_KMAX = 38;       // taps per dimension: down-scaling up to 9x with the bicubic filter's support of 2 (bilinear: 19)
constexpr int PP_BITS = 22;       // PRECISION_BITS of Resample.c for 8-bit channels

// coefficients of output position xx (of `out`) over an input axis of `in` samples: first sample, count, k[] -- precompute_coeffs +
// normalize_coeffs_8bpc; k points into LDS (the weights are evaluated twice instead of being kept in a private array)
__device__ __forceinline__ double pp_weight(double a, int cubic) {
#pragma clang fp contract(off)
  if (a < 0.0) a = -a;
  if (!cubic) return a < 1.0 ? 1.0 - a : 0.0;           // bilinear_filter
  const double A = -0.5;                                  // bicubic_filter (Keys, a = -0.5)
  if (a < 1.0) return ((A + 2.0) * a - (A + 3.0)) * a * a + 1;
  if (a < 2.0) return (((a - 5) * a + 8) * a - 4) * A;
  return 0.0;
}

__device__ __forceinline__ void pp_coeffs(int in, int out, int xx, int cubic, int& xmin, int& cnt, int* k) {
#pragma clang fp contract(off)
  const double scale = (double)in / (double)out;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = (cubic ? 2.0 : 1.0) * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int lo = (int)(center - support + 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)(center + support + 0.5);
  if (hi > in) hi = in;
  hi -= lo;
  if (hi > PP_KMAX) hi = PP_KMAX;   // (never: the host side rejects scales above 9)
  double ww = 0.0;
  for (int x = 0; x < hi; ++x) ww += pp_weight((x + lo - center + 0.5) * ss, cubic);
  for (int x = 0; x < hi; ++x) {
    double v = pp_weight((x + lo - center + 0.5) * ss, cubic);
    if (ww != 0.0) v /= ww;
    k[x] = v < 0.0 ? (int)(-0.5 + v * (double)(1 << PP_BITS)) : (int)(0.5 + v * (double)(1 << PP_BITS));
  }
  xmin = lo;
  cnt = hi;
}

__device__ __forceinline__ int pp_clip8(int v) {
  v >>= PP_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// A workgroup is 64 output columns x 4 output rows of one image: the 64 column and 4 row coefficient sets are computed once (LDS).
// OUT 0: fp32 NCHW, 1: bf16 NHWC (pitch 8), 2: uint8 NHWC before ToTensor
template <int OUT>
__global__ __launch_bounds__(256) void k_image_preprocess(const unsigned char* __restrict__ pool, const ImgDesc* __restrict__ desc, int S,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         void* __restrict__ out, int cubic) {
  __shared__ int s_kx[64][PP_KMAX + 2], s_ky[4][PP_KMAX + 2];   // [..][KMAX] = first sample, [..][KMAX + 1] = count
  const int n = blockIdx.z;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ox = blockIdx.x * 64 + tx;
  const int oy = blockIdx.y * 4 + ty;
  const ImgDesc d = desc[n];
  if (ty == 0 && ox < S) {
    // the flip mirrors the RESIZED image: output column ox shows resized column S - 1 - ox
    const int rx = d.flip ? S - 1 - ox : ox;
    pp_coeffs(d.bw, S, rx, cubic, s_kx[tx][PP_KMAX], s_kx[tx][PP_KMAX + 1], s_kx[tx]);
  }
  if (tx == 0 && oy < S) pp_coeffs(d.bh, S, oy, cubic, s_ky[ty][PP_KMAX], s_ky[ty][PP_KMAX + 1], s_ky[ty]);
  __syncthreads();
  if (ox >= S || oy >= S) return;
  const int xmin = s_kx[tx][PP_KMAX], xn = s_kx[tx][PP_KMAX + 1], ymin = s_ky[ty][PP_KMAX], yn = s_ky[ty][PP_KMAX + 1];
  const unsigned char* base = pool + d.off + ((long)(d.bi + ymin) * d.W + (d.bj + xmin)) * 3;
  const long pitch = (long)d.W * 3;
  int v0 = 1 << (PP_BITS - 1), v1 = v0, v2 = v0;
  for (int y = 0; y < yn; ++y) {
    const unsigned char* row = base + y * pitch;
    int h0 = 1 << (PP_BITS - 1), h1 = h0, h2 = h0;
    for (int x = 0; x < xn; ++x) {
      const int kk = s_kx[tx][x];
      h0 += row[3 * x + 0] * kk;
      h1 += row[3 * x + 1] * kk;
      h2 += row[3 * x + 2] * kk;
    }
    // the horizontal pass is rounded to uint8 before the vertical one (two-pass resampling through an 8-bit image)
    const int kk = s_ky[ty][y];
    v0 += pp_clip8(h0) * kk;
    v1 += pp_clip8(h1) * kk;
    v2 += pp_clip8(h2) * kk;
  }
  const int p0 = pp_clip8(v0), p1 = pp_clip8(v1), p2 = pp_clip8(v2);
  if (OUT == 2) {
    unsigned char* o = reinterpret_cast<unsigned char*>(out) + (((long)n * S + oy) * S + ox) * 3;
    o[0] = (unsigned char)p0; o[1] = (unsigned char)p1; o[2] = (unsigned char)p2;
    return;
  }
  // ToTensor (u8 -> fp32 / 255), Normalize ((t - mean) / std): fp32, correctly rounded division
  const float f0 = ((float)p0 / 255.0f - m0) / s0, f1 = ((float)p1 / 255.0f - m1) / s1, f2 = ((float)p2 / 255.0f - m2) / s2;
  if (OUT == 1) {
    bf16_t* o = reinterpret_cast<bf16_t*>(out) + (((long)n * S + oy) * S + ox) * 8;   // channel pitch 8 (zero padding)
    bf16x8 v;
    v[0] = (bf16_t)f0; v[1] = (bf16_t)f1; v[2] = (bf16_t)f2;
#pragma unroll
    for (int e = 3; e < 8; ++e) v[e] = (bf16_t)0.f;
    *reinterpret_cast<bf16x8*>(o) = v;
  } else {
    float* o = reinterpret_cast<float*>(out) + (long)n * 3 * S * S + (long)oy * S + ox;
    o[0] = f0;
    o[(long)S * S] = f1;
    o[2L * S * S] = f2;
  }
}

}  // namespace atomnas

using namespace atomnas;

// include/atomnas_hip.h: pool = the packed uint8 HWC images, desc = device array of N atomnas_img_desc (ImgDesc above + 4 bytes of padding),
// out_mode 0: fp32 NCHW, 1: bf16 NHWC (channel pitch 8), 2: uint8 [N][S][S][3] before ToTensor (parity against PIL); filter 0: PIL's BILINEAR,
// 1: PIL's BICUBIC resampler.
extern "C" int atomnas_image_preprocess(const void* pool, const void* desc, int N, int S, const float* mean3, const float* std3, void* out,
                                        int out_mode, int filter, void* stream) {
  ATOMNAS_REQUIRE(pool && desc && out && N > 0 && S > 0 && S <= 1024, "image_preprocess: bad arguments");
  ATOMNAS_REQUIRE(out_mode == 2 || (mean3 && std3), "image_preprocess: mean / std (host arrays of 3 floats) are required");
  ATOMNAS_REQUIRE(out_mode >= 0 && out_mode <= 2, "image_preprocess: out_mode %d", out_mode);
  ATOMNAS_REQUIRE(filter == 0 || filter == 1, "image_preprocess: filter %d (0 = PIL BILINEAR, 1 = PIL BICUBIC)", filter);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((S + 63) / 64, (S + 3) / 4, N), block(256);
  const ImgDesc* d = reinterpret_cast<const ImgDesc*>(desc);
  const float m0 = mean3 ? mean3[0] : 0.f, m1 = mean3 ? mean3[1] : 0.f, m2 = mean3 ? mean3[2] : 0.f;
  const float s0 = std3 ? std3[0] : 1.f, s1 = std3 ? std3[1] : 1.f, s2 = std3 ? std3[2] : 1.f;
  if (out_mode == 2) hipLaunchKernelGGL(k_image_preprocess<2>, grid, block, 0, st, (const unsigned char*)pool, d, S, m0, m1, m2, s0, s1, s2, out, filter);
  else if (out_mode == 1) hipLaunchKernelGGL(k_image_preprocess<1>, grid, block, 0, st, (const unsigned char*)pool, d, S, m0, m1, m2, s0, s1, s2, out, filter);
  else hipLaunchKernelGGL(k_image_preprocess<0>, grid, block, 0, st, (const unsigned char*)pool, d, S, m0, m1, m2, s0, s1, s2, out, filter);
  return check_launch("image_preprocess");
}

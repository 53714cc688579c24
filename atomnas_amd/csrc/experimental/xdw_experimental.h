/* Round-4 experiment, NOT part of the product ABI: the expand 1x1 convolution + BatchNorm + activation recomputed on chip inside the
 * depthwise kernels ("E-elimination": the expanded tensor never in HBM).  Built by tools/build_xdw_experiment.sh into
 * atomnas_amd/csrc/build/variants/libxdw.so; measured slower than the stand-alone kernels (profiles/r04_xdw_*). */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* ---- E-elimination (csrc/experimental/xdw_fused.hip + k_xdwb in csrc/dwconv_cw.hip; bf16, stride 1, inp <= 64): the expand 1x1 convolution, its BatchNorm and activation
 *      (models/mobilenet_base.py:316-320) computed on chip in front of the depthwise convolution (:330-336), forward and backward.
 *      The expanded tensor E = x We^T never exists in HBM; x: the block input [M = N*H*W][ldx] (plain layout).
 * atomnas_xdw_supported: 1 when the three entry points below have instances for a branch segment of C hidden channels.
 * atomnas_gram: G = X^T X [inp][inp] and sx = sum_m x_m [inp] of the block input (one pass over the narrow tensor; per-workgroup
 *   partials of inp*inp + inp floats in the caller's workspace ws, summed in workgroup order).
 * atomnas_gram_stats: statistics row of the expand BatchNorm from them:
 *     stats[c] = sum_m e_c = w_c . sx,   stats[stat_ld + c] = sum_m e_c^2 = w_c^T G w_c      (one row: stat_rows = 1 for the finalize)
 *   wexp: packed expand weight [C][ldwe] (atomnas_pack_weights mode 0).
 * atomnas_xdw_fwd:  y = dwconv_k(act(in_scale * (x wexp^T) + in_shift)) of one branch segment (wexp / in_scale / in_shift / w / y
 *   point at the segment), statistics rows [sum y, sum y^2] as atomnas_dwconv_fwd.
 * atomnas_xdw_bwd:  atomnas_dwconv_bwd with e = x wexp^T (fp32, recomputed) in place of its input stream: h, dw, stats [sum h, sum h*e].
 * atomnas_xb_coeffs: with dE = c1*h + c2*E + c3 the expand backward is
 *     dX = (c1*h) We + x M + v,   dWe = diag(c1) h^T x + diag(c2) We G + c3 sx^T,     M = We^T diag(c2) We,  v = c3^T We
 *   this entry writes mp = bf16(M) in atomnas_pw_gemm_nt's weight layout ([inp rounded up to 64][ldm], padding zeroed by the caller),
 *   vb = v (its bias) and adds the last two terms to dwe[C*inp]; the h terms are atomnas_expand_bwd (e = NULL) or
 *   atomnas_pw_gemm_nt / atomnas_pw_gemm_tn with c1 as their BNRELU scale. */
int atomnas_xdw_supported(int N, int H, int W, int inp, int C, int k, int stride, int dtype);
int atomnas_gram_stats(const float* gram, int ldg, const float* sx, const void* wexp, int ldwe, int inp, int C, float* stats, int stat_ld,
                       void* stream);
int atomnas_xdw_fwd(const void* x, int ldx, int inp, const void* wexp, int ldwe, const float* in_scale, const float* in_shift, int act,
                    const float* w, int ldw, void* y, long y_ss, float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k,
                    int dtype, void* stream);
int atomnas_xdw_bwd(const void* g, long g_ss, const void* yraw, long yraw_ss, const float* c1, const float* c2, const float* c3, const void* x,
                    int ldx, int inp, const void* wexp, int ldwe, const float* in_scale, const float* in_shift, int act, const float* w, int ldw,
                    void* h, long h_ss, float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k,
                    int dtype, void* stream);

#ifdef __cplusplus
}
#endif

/*
 * atomnas_hip.h -- C ABI of libatomnas_hip.so, the gfx950 (MI355X) kernels behind the AtomNAS supernet-training hot path.
 *
 * The reference (meijieru/AtomNAS) has no native code: every FLOP of this path runs inside ATen.  Each entry point
 * below therefore replaces the ATen call(s) that a reference Python line makes; the line is cited as
 * <reference file>:<line>.  A reference maintainer binds these with ctypes.
 *
 * Conventions
 *   - plain C: device pointers, ints, floats; no torch / C++ types.  `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success, non-zero on failure; atomnas_last_error() returns the message.
 *   - nothing is allocated or freed by the library; workspaces are caller-owned.  Functions are asynchronous on `stream`,
 *     re-entrant per stream, and contain no host synchronisation (they can be captured into a hipGraph).
 *   - activations are NHWC viewed as [M = N*H*W rows][C channels].  Every activation argument comes as (pointer, ld, ss):
 *       ss == 0  plain layout: element (row, c) at row*ld + c; ld is a multiple of 8 and channels C..ld-1 hold zeros;
 *       ss  > 0  slab-major layout: the channels are cut into slabs of 16 and each slab is a contiguous [M][16] matrix,
 *                element (row, c) at (c/16)*ss + row*16 + c%16, ss >= M*16 (ld is ignored).  This is the layout of the
 *                6x-expanded hidden tensors of a block: a workgroup that owns a channel range streams contiguous memory.
 *     dtype: 0 = fp32, 1 = bf16 storage (fp32 accumulation).
 *   - per-channel fp32 vectors (scale, shift, c1..c3, gamma, ...) are readable up to C rounded up to 8.
 *   - "stats" outputs are PARTIAL ROWS: an fp32 buffer [stat_rows][2][pitch] (pitch = stat_ld, or N for the GEMM).  The
 *     producing function writes EVERY row of its channel range with plain stores (one workgroup or wave owns a row; rows it
 *     does not need are zero-filled), so the buffer needs no initialisation, nothing is accumulated atomically, and the
 *     BatchNorm finalize functions, which sum rows 0..stat_rows-1 in a fixed order, give bit-identical results run to run.
 *     More rows allow more concurrent workgroups (ATOMNAS_STAT_ROWS_DEFAULT is a good value for 64 <= C < 1024).
 *   - weight gradients (depthwise taps, 1x1 weights) are likewise reduced from per-workgroup partials in a caller-owned
 *     workspace in a fixed order; no entry point of this library accumulates floating-point values with atomics.
 */
#ifndef ATOMNAS_HIP_H
#define ATOMNAS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ATOMNAS_STAT_ROWS_DEFAULT 512

#define ATOMNAS_DT_F32 0
#define ATOMNAS_DT_BF16 1

/* activation mode carried by every `*_relu`, `relu` and `mask` argument: models/mobilenet_base.py:407-415 get_active_fn */
#define ATOMNAS_ACT_NONE 0
#define ATOMNAS_ACT_RELU 1  /* max(a, 0);           backward passes where a > 0       */
#define ATOMNAS_ACT_RELU6 2 /* min(max(a, 0), 6);   backward passes where 0 < a < 6   */
#define ATOMNAS_ACT_SWISH 3 /* a * sigmoid(a);      backward multiplies by s*(1 + a*(1-s))   models/mobilenet_base.py:72-80 */

/* prologue applied to a GEMM operand while it is loaded */
#define ATOMNAS_PRO_NONE 0   /* a                                   */
#define ATOMNAS_PRO_BNRELU 1 /* act(a * c1[k] + c2[k])              BatchNorm apply (+ReLU) of the producer */
#define ATOMNAS_PRO_BNBWD 2  /* c1[k]*a + c2[k]*a2 + c3[k]          BatchNorm backward, a = masked grad, a2 = raw activation */

/* statistics taken in a GEMM epilogue */
#define ATOMNAS_STAT_NONE 0
#define ATOMNAS_STAT_SQ 1 /* [sum c, sum c^2]   -> forward BatchNorm of the output           */
#define ATOMNAS_STAT_Z 2  /* [sum c, sum c*z]   -> backward BatchNorm of the tensor z        */

/* indices into the device-resident hyper-parameter vector (float[4]) */
#define ATOMNAS_HYP_LR 0
#define ATOMNAS_HYP_RHO 1
#define ATOMNAS_HYP_EMA_DECAY 2
#define ATOMNAS_HYP_GRAD_SCALE 3

const char* atomnas_last_error(void);
#define ATOMNAS_ABI_VERSION 9   /* 2: (ptr, ld, ss) activations, partial-row statistics, workspaces, SE / fused-backward entry points; 3: + atomnas_bnbwd_apply; 4: + atomnas_gram, atomnas_xb_coeffs, atomnas_expand_bwd with e = NULL; 5: the SE dense layers take their weights packed over the padded channel layout; 6: + atomnas_dwconv_mm_supported; 7: + atomnas_image_preprocess; 8: + atomnas_gather_jobs; 9: atomnas_expand_bwd / atomnas_project_bwd lose their two-stream forms (arguments e, c2, c3 / p, c1, c2, c3), + atomnas_fold_jobs, atomnas_image_preprocess takes the resampling filter */
int atomnas_abi_version(void);
int atomnas_runtime_version(void);

/* ---- depthwise k x k convolution: nn.Conv2d(C, C, k, stride, (k-1)/2, groups=C, bias=False)
 *      models/mobilenet_base.py:330-336 (built through ConvBNReLU :120-142); k in {3,5,7}, stride in {1,2}.
 * forward: y = dwconv(act(x*in_scale+in_shift));  stats rows [sum y, sum y^2] for channels 0..C-1   (in_scale == NULL: x as is)
 *   w: fp32 taps [k*k][ldw] (tap-major, see atomnas_pack_weights mode 2). */
int atomnas_dwconv_fwd(const void* x, int ldx, long x_ss, const float* in_scale, const float* in_shift, int in_relu, const float* w,
                       int ldw, void* y, int ldy, long y_ss, float* stats, int stat_ld, int stat_rows, int N, int H, int W, int C, int k, int stride,
                       int dtype, void* stream);

/* backward (input gradient and weight gradient in one pass over the data):
 *   dYraw = c1*g + c2*yraw + c3  (yraw == NULL: dYraw = g)      -- BatchNorm backward of the BN after the conv
 *   h     = dwconv^T(dYraw) * [x*in_scale+in_shift > 0]         -- ReLU backward of the producer (if in_relu)
 *   dw[c][k*k] += corr(act(x*in_scale+in_shift), dYraw)         -- torch layout [C,1,k,k], fp32; per-workgroup partials go to
 *                                                                   dw_ws [part_rows][C][k*k] and are summed in a fixed order
 *   stats rows [sum h, sum h*x]                                 -- for the producer's BatchNorm backward ([part_rows][2][stat_ld])
 *   part_rows bounds the number of workgroups per channel slab (each owns one row of stats and of dw_ws). */
int atomnas_dwconv_bwd(const void* g, int ldg, long g_ss, const void* yraw, int ldyr, long yraw_ss, const float* c1, const float* c2,
                       const float* c3, const void* x, int ldx, long x_ss, const float* in_scale, const float* in_shift, int in_relu,
                       const float* w, int ldw, void* h, int ldh, long h_ss, float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W,
                       int C, int k, int stride, int dtype, void* stream);

/* 1 when the two entry points above run the stride-1 / slab-major "channel pair per wave" kernels (csrc/dwconv_cw.hip) for this
 *   shape with slab-major tensors, 0 when they run the tile kernels (csrc/dwconv.hip); dir: 0 forward, 1 backward.  Same results
 *   either way; a query for tests and launch-geometry tools. */
int atomnas_dwconv_cw_supported(int N, int H, int W, int C, int k, int stride, int dtype, int dir);

/* 1 when they run the matrix-core kernels (csrc/dwconv_mm.hip: bf16, stride 1, slab-major tensors; csrc/dwconv_mm2.hip: the stride-2
 *   forward; the tap arithmetic as MFMAs against a Toeplitz operand of the taps) for this shape; dir: 0 forward, 1 backward.  Those kernels round the MFMA operands
 *   (forward: activated input and taps to fp16; backward: the gradient of the raw output, the taps and the activated input to bf16),
 *   which oracle/atomnas_oracle.py restates where this query says so (bf16_storage_mm). */
int atomnas_dwconv_mm_supported(int N, int H, int W, int C, int k, int stride, int dtype, int dir);

/* ---- pointwise (1x1) convolutions as MFMA GEMMs: models/mobilenet_base.py:316-320 (expand), :338 (project),
 *      models/mobilenet_supernet.py:148-153 (last conv), :160-163 (classifier); branches concatenated (:378).
 * C[M,N] = epilogue( prologue(A)[M,K] x Wp[N,K]^T )
 *   prologue: a_mode (ATOMNAS_PRO_*), second stream a2, coefficient vectors ac1..ac3, a_relu for BNRELU
 *   wp: packed weights, storage dtype, [N rounded up to 64][ldw], ldw >= K rounded up to 32 (bf16) / 4 (fp32), padding zero
 *   epilogue: + bias[n]; + add[m][n]; if mask: c = 0 where z*zscale+zshift <= 0; store (fp32 if out_f32);
 *             stats per stat_mode on the stored value: partial rows [stat_rows][2][N]. */
int atomnas_pw_gemm_nt(int a_mode, const void* a, int lda, long a_ss, const void* a2, int lda2, long a2_ss, const float* ac1,
                       const float* ac2, const float* ac3, int a_relu, const void* wp, int ldw, void* c, int ldc, long c_ss, int out_f32,
                       const void* add, int ldadd, const void* z, int ldz, long z_ss, const float* zscale, const float* zshift, int mask, const float* bias,
                       float* stats, int stat_mode, int stat_rows, long M, int N, int K, int dtype, void* stream);

/* weight gradient: out[i*si + j*sj] += sum_m prologue(U)[m,i] * prologue(V)[m,j]  (fp32).  The reduction over M is split into
 *   row chunks whose partial outputs [chunk][NU][NV] go to the caller-owned workspace ws (ws_floats floats) and are summed in
 *   chunk order; ws == NULL (or room for < 2 partials): one workgroup per output tile walks all of M. */
int atomnas_pw_gemm_tn(int u_mode, const void* u, int ldu, long u_ss, const void* u2, int ldu2, long u2_ss, const float* uc1,
                       const float* uc2, const float* uc3, int u_relu, int NU, int v_mode, const void* v, int ldv, long v_ss,
                       const void* v2, int ldv2, long v2_ss, const float* vc1, const float* vc2, const float* vc3, int v_relu, int NV, float* out, long si, long sj, long M,
                       float* ws, long ws_floats, int dtype, void* stream);

/* Backward of the expand convolution ConvBNReLU(inp, hid, 1) (models/mobilenet_base.py:316-320) WITHOUT its raw output E, both gradients
 *   from ONE pass over the masked hidden gradient h (bf16 storage; shapes per atomnas_expand_bwd_supported: the early, activation-dominated
 *   stages).  With the BatchNorm backward dE = c1*h + c2*E + c3 and E = x We^T (x: the block input [M][ldx], plain layout):
 *       dX = (c1*h) We + x M + v,   dWe = diag(c1) h^T x + diag(c2) We G + c3 sx^T,     M = We^T diag(c2) We,  v = c3^T We,
 *       G = X^T X,  sx = sum_m x_m
 *   the c2 / c3 terms are inp x inp sized, so the wide tensors are read once and E not at all (csrc/xbwd.hip):
 * atomnas_gram: G [inp][inp] and sx [inp] from one pass over x (inp <= 64 and a multiple of 8); per-workgroup partials of inp*inp + inp
 *   floats in the caller's workspace ws (ws_floats floats, at least one partial), summed in workgroup order.
 * atomnas_xb_coeffs: writes mp = bf16(M) in atomnas_pw_gemm_nt's weight layout ([inp rounded up to 64][ldm], padding zeroed by the
 *   caller), vb = v (its bias) and adds the last two terms of dWe to dwe[C*inp]; wexp: packed expand weight [C][ldwe]
 *   (atomnas_pack_weights mode 0); c2 / c3: coefficients of the C hidden channels.
 * atomnas_expand_bwd (ABI 9: the two-stream form that also read E is gone; the arguments e, c2, c3 with it):
 *     gx[M, inp] = (c1*h) * We (+ x mp^T + vb when mp != NULL) (+ add: the residual branch);   dwe[n*inp + k] += sum_m c1[n]*h[m][n] * x[m][k]
 *   wt = We^T packed by atomnas_pack_weights ([inp padded to 64][ldw >= hid rounded up to 32]); ws: ws_floats floats for the
 *   per-workgroup partials of the weight gradient (inp*hid floats each; summed in workgroup order).  Replaces one atomnas_pw_gemm_nt
 *   (BNBWD prologue) + one atomnas_pw_gemm_tn, which read h and E twice.  Wider layers: the same h terms through atomnas_pw_gemm_nt /
 *   atomnas_pw_gemm_tn with c1 as their BNRELU scale. */
int atomnas_expand_bwd_supported(int inp, int hid, int dtype);
int atomnas_expand_bwd(const void* h, int ldh, long h_ss, const float* c1, const void* x, int ldx, const void* wt, int ldw, const void* add,
                       int ldadd, void* gx, int ldgx, float* dwe, float* ws, long ws_floats, const void* mp, int ldm, const float* vb, long M,
                       int inp, int hid, int dtype, void* stream);
int atomnas_gram(const void* x, int ldx, long M, int inp, float* ws, long ws_floats, float* gram, float* sx, int dtype, void* stream);
int atomnas_xb_coeffs(const float* c2, const float* c3, const void* wexp, int ldwe, const float* gram, int ldg, const float* sx, int inp,
                      int C, void* mp, int ldm, float* vb, float* dwe, void* stream);

/* Backward of the linear projection nn.Conv2d(hid, oup, 1) (models/mobilenet_base.py:338) in ONE pass over the raw depthwise
 *   output z (bf16 storage; shapes per atomnas_project_bwd_supported: oup a multiple of 8 and <= 64; layouts per
 *   atomnas_project_bwd_dp_supported: hidden tensors slab-major, or plain with hid % 8 == 0):
 *     g = dP, the differentiated block-output BatchNorm as a tensor [M, oup] (atomnas_bnbwd_apply's output; ABI 9: the form that computed
 *       c1*g + c2*p + c3 per tile from two streams is gone, the arguments p, c1, c2, c3 with it)
 *     gh[M, hid] = act'(z*zscale + zshift) * (dP * Wp),  statistics rows [sum gh, sum gh*z]
 *     dwp[o*si + n*sj] += sum_m dP[m][o] * act(z*zscale + zshift)[m][n]
 *   wpt = Wp^T packed by atomnas_pack_weights ([hid padded to 64][ldw >= oup rounded up to 32]); ws: per-row-range partials of the
 *   weight gradient (oup*hid floats each).  Replaces atomnas_pw_gemm_nt(mask, STAT_Z) + atomnas_pw_gemm_tn on dP. */
int atomnas_project_bwd_supported(int oup, int hid, int dtype);
int atomnas_project_bwd_dp_supported(long M, int oup, int hid, int ldg, int ldz, long z_ss, int ldgh, long gh_ss, int stat_rows, int dtype);
int atomnas_project_bwd(const void* g, int ldg, const void* wpt, int ldw, const void* z, int ldz, long z_ss, const float* zscale,
                        const float* zshift, int act, void* gh, int ldgh, long gh_ss, float* stats, int stat_rows, float* dwp, long si,
                        long sj, float* ws, long ws_floats, long M, int oup, int hid, int dtype, void* stream);

/* ---- BatchNorm2d (training, eval and cumulative-calibration modes): models/mobilenet_base.py:142,342;
 *      kwargs from models/mobilenet_supernet.py:95-98; calibration mode utils/common.py:214-226.
 * finalize forward: stats = stat_rows partial rows [2][stat_ld] of [sum x, sum x^2] over `count` elements (stat_ld >= C rounded
 *   up to 8: a branch segment of a wider statistics buffer can be finalized on its own) -> scale = gamma*invstd,
 *   shift = beta - mean*scale, save_mean / save_invstd for backward, running statistics update (momentum < 0: cumulative
 *   average 1/(counter+1); the caller bumps the counter). */
int atomnas_bn_finalize_fwd(const float* stats, int stat_rows, int stat_ld, double count, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* scale, float* shift,
                            float* save_mean, float* save_invstd, int C, const int* cmap, void* stream);
/* ABI 5, cmap (may be NULL = identity), all three entry points: the statistics and coefficient vectors are indexed by the kernel
 *   channel c < C (a fused block's padded branch segments), the module's parameter / running-statistics / gradient vectors by
 *   cmap[c]; cmap[c] = -1 marks padding inside the range, whose coefficients are written as zeros.  One launch then covers the
 *   whole padded width of a fused block's expand BatchNorm (models/mobilenet_base.py:236-247) instead of one per branch segment. */
/* eval mode: scale/shift from the running statistics */
int atomnas_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                           float* scale, float* shift, int C, const int* cmap, void* stream);
/* finalize backward: stats2=[sum g, sum g*x] -> dgamma (+ rho*penalty*sign(gamma), utils/prune.py:161-167), dbeta and the
 *   coefficients of dx = c1*g + c2*x + c3.  rho is read from device memory (rho_ptr, may be NULL). */
int atomnas_bn_finalize_bwd(const float* stats2, int stat_rows, int stat_ld, double count, const float* gamma, const float* save_mean, const float* save_invstd,
                            const float* rho_ptr, const float* penalty, float* dgamma, float* dbeta, float* c1, float* c2, float* c3,
                            int C, const int* cmap, void* stream);
/* y = act(x*scale+shift) (+ res): the shared pw_bn + residual of a block, models/mobilenet_base.py:379-381 */
int atomnas_bn_apply(const void* x, int ldx, const float* scale, const float* shift, int relu, const void* res, int ldres, void* y,
                     int ldy, long M, int C, int dtype, void* stream);
/* y = c1*g + c2*x + c3 per channel (plain layouts): the gradient through a training-mode BatchNorm as a tensor, i.e. what the
 *   PRO_BNBWD prologue of the GEMMs computes per tile (torch.nn.BatchNorm2d backward behind models/mobilenet_base.py:338-339).
 *   Used where one narrow gradient feeds many GEMM tiles (ABI 3).  Whole 8-channel groups are moved: c1, c2, c3 must be readable (any
 *   finite value) up to C rounded up to 8, and g, x, y, c1, c2, c3 16-byte aligned. */
int atomnas_bnbwd_apply(const void* g, int ldg, const void* x, int ldx, const float* c1, const float* c2, const float* c3, void* y,
                        int ldy, long M, int C, int dtype, void* stream);
/* pooled[n][c] = dropout(mean_hw act(x*scale+shift)): last ConvBNReLU activation + AvgPool2d + Dropout,
 *   models/mobilenet_supernet.py:148-163 (keep mask written for backward; step_ptr decorrelates iterations) */
int atomnas_bn_act_pool(const void* x, int ldx, const float* scale, const float* shift, int relu, void* pooled, int ldp,
                        unsigned char* keep, float drop_p, unsigned long long seed, const long long* step_ptr, int N, int HW, int C,
                        int dtype, void* stream);
int atomnas_pool_act_bwd(const void* dpooled, int ldp, const unsigned char* keep, float drop_p, const void* x, int ldx,
                         const float* scale, const float* shift, int relu, void* g, int ldg, float* stats2, int stat_rows, int N,
                         int HW, int C, int dtype, void* stream);
/* g = dy * [z*scale+shift > 0] (scale == NULL: no mask);  stats2 rows [sum g, sum g*z];  g may be NULL (statistics only) */
int atomnas_act_bwd_stats(const void* dy, int lddy, const void* z, int ldz, const float* scale, const float* shift, int relu, void* g,
                          int ldg, float* stats2, int stat_rows, long M, int C, int dtype, void* stream);

/* ---- Squeeze-and-Excitation of the fused block (AtomNAS+): models/mobilenet_base.py:93-117 inside :256-267.
 *   A = act(D*scale+shift) is the activated depthwise output (never materialised), D the raw depthwise output [M = N*HW][C];
 *   cmap[c]: row / column of the reference's SE weights (w1 [hid][total], w2 [total][hid]) for padded channel c, -1 for padding;
 *   ABI 5: the kernels read fp32 copies of the weights packed over the block's padded channels, channels contiguous in both:
 *   w1p[j][c] = w1[j][cmap c], w2t[j][c] = w2[cmap c][j], b2p[c] = b2[cmap c], zeros at padding (atomnas_pack_weights jobs);
 *   the weight GRADIENTS are written in the reference's layouts through cmap;
 *   pooled / gate / dgate / dz2 / dpooled: fp32 [N][ldg];  hpre / dz1: fp32 [N][hid].
 * forward:  pooled = mean_hw A;  hpre = w1*pooled + b1;  gate = sigmoid(w2*act(hpre) + b2);  out = A * gate[n] */
/* ABI 5: the per-image sums leave atomnas_se_squeeze (and the dgate pass inside atomnas_se_bwd_gate) as `parts` planes
 *   [parts][N][ld] (plane pitch part_stride floats), each over a share of the image's pixels, parts = atomnas_se_pool_parts(N, HW, C)
 *   in 1..16; the dense-layer kernels add the planes in plane order and atomnas_se_mlp_fwd stores the sum in `pooled`. */
int atomnas_se_pool_parts(int N, int HW, int C);
int atomnas_se_squeeze(const void* d, int ldd, long d_ss, const float* scale, const float* shift, int act, float* pooled_parts, int ldp,
                       int parts, long part_stride, int N, int HW, int C, int dtype, void* stream);
int atomnas_se_mlp_fwd(const float* pooled_parts, int ldp, int parts, long part_stride, float* pooled, const int* cmap, const float* w1p,
                       const float* b1, const float* w2t, const float* b2p, int act, float* hpre, float* gate, int N, int HT, int hid,
                       void* stream);
int atomnas_se_scale(const void* d, int ldd, long d_ss, const float* scale, const float* shift, int act, const float* gate, int ldg,
                     void* out, int ldo, long o_ss, long M, int HW, int C, int dtype, void* stream);
/* backward of the gate: dgate = sum_hw dS*A;  dz2 = dgate*gate*(1-gate);  dz1 = act'(hpre) * w2^T dz2;  dpooled = w1^T dz1;
 *   dw1 += dz1^T pooled, db1 += sum dz1, dw2 += dz2^T act(hpre), db2 += sum dz2 (batch loops in image order: bit-reproducible);
 *   act: the block's activation (A), se_act: the activation between the two dense layers (ABI 5: separate arguments) */
int atomnas_se_bwd_gate(const void* ds, int ldds, long ds_ss, const void* d, int ldd, long d_ss, const float* scale, const float* shift,
                        int act, const float* gate, const float* pooled, int ldg, const int* cmap, const float* w1p, const float* w2t,
                        const float* hpre, float* dgate, int parts, long part_stride, float* dz2, float* dz1, float* dpooled, float* dw1,
                        float* db1, float* dw2,
                        float* db2, int se_act, int N, int HW, int HT, int total, int hid, int dtype, void* stream);
/* g = act'(D*scale+shift) * (dS*gate[n] + dpooled[n]/HW): the gradient wrt the depthwise BatchNorm output;  stats2 rows [sum g, sum g*D] */
int atomnas_se_bwd_apply(const void* ds, int ldds, long ds_ss, const void* d, int ldd, long d_ss, const float* scale, const float* shift,
                         int act, const float* gate, const float* dpooled, int ldg, void* g, int ldgo, long g_ss, float* stats2,
                         int stat_rows, long M, int HW, int C, int dtype, void* stream);

/* ---- stem and loss
 * im2col of the 3x3 stride-2 stem conv (models/mobilenet_supernet.py:126-132): img NCHW fp32 -> col [N*Ho*Wo][ld>=32] */
int atomnas_im2col_stem(const float* img, void* col, int ld, int N, int H, int W, int dtype, void* stream);
/* CrossEntropyLabelSmooth (utils/optim.py:180-207) + top-1/top-5 hit counters (common.py:73-79), no host sync.
 * dlogits = d(mean loss)/dlogits * gscale, storage dtype, columns K..ldd-1 zeroed. */
int atomnas_ce_smooth(const float* logits, int ldl, const long long* target, float eps, int B, int K, float* loss_per_sample,
                      void* dlogits, int ldd, float gscale, int* topk_correct, int dtype, void* stream);
int atomnas_colsum(const void* x, int ld, float* out, long M, int C, int dtype, void* stream);

/* ---- optimizer tail on flat fp32 arenas (one launch for all parameters)
 * L2 'mnas' (utils/optim.py:226-243) + RMSprop.step (utils/rmsprop.py:70-132) + EMA (utils/optim.py:54-65):
 *   g' = g*hyper[GRAD_SCALE] + wd_chunk[i/256]*p;  sq = alpha*sq + (1-alpha)*g'^2;  avg = sqrt(sq+eps) | sqrt(sq)+eps;
 *   buf = momentum*buf + g'/avg;  p -= hyper[LR]*buf;  ema = d*ema + (1-d)*p with d = hyper[EMA_DECAY] (d < 0: skip).
 *   l2_value (optional): l2_value[0] = 0.5 * sum_i wd_chunk[i/256] * p_i^2 at the weights BEFORE the update (the value of
 *   cal_l2_loss for logging, train.py:206-208), summed in a fixed order through the 4096-float workspace ws. */
int atomnas_fused_rmsprop_ema(float* p, const float* g, float* sq, float* buf, float* ema, const float* wd_chunk, long n,
                              const float* hyper, double alpha, double eps, int eps_inside_sqrt, double momentum, float* l2_value,
                              float* ws, void* stream);
/* out[0] = scale * sum_i x[i], fixed summation order: mean of the per-sample losses (train.py:178-180) */
int atomnas_vec_sum(const float* x, int n, float scale, float* out, void* stream);
int atomnas_ema_update(float* shadow, const float* x, long n, const float* hyper, void* stream);
/* x[i] *= hyper[idx]: the BN running statistics summed over the ranks become their average (utils/distributed.py:164-169,
 * allreduce_bn; hyper[3] = 1 / world) */
int atomnas_scale_by(float* x, long n, const float* hyper, int idx, void* stream);
/* housekeeping of the step without framework kernels: zero-fill (16-byte aligned, whole words), int64 counters += v
 * (num_batches_tracked of the BatchNorms, the dropout step counter) */
int atomnas_zero(void* p, long bytes, void* stream);
int atomnas_add_i64(long* p, long n, long v, void* stream);
/* regularisers as gradient contributions / values over a job table {long off; int count; float coef;}:
 *   cal_l2_loss (utils/optim.py:210-249): g += wd*p, value 0.5*wd*sum p^2;  cal_bn_l1_loss (utils/prune.py:161-167):
 *   g += rho*penalty*sign(gamma), value rho*penalty*sum|gamma|.  mult_ptr / grad_out_ptr: optional device scalars. */
int atomnas_reg_grad(const float* p, float* g, const void* jobs_dev, int njobs, int use_sign, const float* mult_ptr,
                     const float* grad_out_ptr, void* stream);
int atomnas_reg_value(const float* p, const void* jobs_dev, int njobs, int use_abs, const float* mult_ptr, float post_scale,
                      float* out, float* ws /* scratch, 64 * njobs floats */, void* stream);
/* re-pack fp32 master weights into kernel layouts; jobs_dev: device array of
 *   struct { long src_off, dst_off; int rows, cols, src_ld, dst_ld, c_off, mode; }  (mode 0 [N][K], 1 transposed, 2 depthwise taps) */
int atomnas_pack_weights(const float* arena, void* packbuf, const void* jobs_dev, int njobs, int dtype, void* stream);

/* ---- input pipeline (utils/dataflow.py:92-170 'imagenet1k_mnas_bilinear', utils/transforms.py:54-177): decoded uint8
 *      HWC images -> crop -> PIL-exact resize (filter 0: BILINEAR, antialiased triangle filter; filter 1: BICUBIC, Keys cubic a = -0.5
 *      -- 'imagenet1k_mnas_bicubic', the reference's default; 22-bit coefficients, horizontal pass rounded to uint8, then vertical:
 *      libImaging/Resample.c) -> horizontal flip -> ToTensor -> Normalize, one launch per batch.
 * pool: the images, packed (3 channels, row pitch 3 W);  desc: device array of N atomnas_img_desc;  S: output side;
 * mean3 / std3: HOST arrays of three floats (read at the call);  out_mode 0: fp32 NCHW [N][3][S][S] (what atomnas_im2col_stem reads),
 * 1: bf16 NHWC with a channel pitch of 8 (padding zero), 2: uint8 [N][S][S][3], the resized and flipped image before ToTensor
 * (parity against PIL).  The crop box must lie inside the image and be at most 9 S on a side (host-side check: utils/dataflow.py). */
typedef struct atomnas_img_desc {
  long off;                 /* byte offset of the image in the pool */
  int H, W;                 /* decoded size */
  int bi, bj, bh, bw;       /* crop box: top, left, height, width */
  int flip;                 /* mirror the resized image horizontally */
  int pad_;
} atomnas_img_desc;
int atomnas_image_preprocess(const void* pool, const void* desc, int N, int S, const float* mean3, const float* std3, void* out,
                             int out_mode, int filter, void* stream);

/* ---- deferred fixed-order reductions (ABI 5).  The weight-gradient entry points (atomnas_pw_gemm_tn, atomnas_dwconv_bwd,
 *   atomnas_expand_bwd, atomnas_project_bwd) write per-workgroup partials to their workspace and sum them in a fixed order with one
 *   small launch each: ~110 launches of a few microseconds per supernet step, every one a ~5 us node of the step's hipGraph.
 *   atomnas_reduce_defer(1): from now on those sums are only RECORDED (process-wide; backward may run on another thread);
 *   atomnas_reduce_flush(stream): sums all recorded jobs with one launch per 56 jobs (the job table travels in the kernel arguments,
 *   so the launch is capturable), same per-element order of additions: bit-identical results.  The caller keeps the workspaces of
 *   the recorded calls alive and unmodified until the flush, and flushes before anything reads the gradients (collective,
 *   optimizer).  atomnas_reduce_defer(0) flushes what is recorded and returns to immediate reductions. */
int atomnas_reduce_defer(int on, void* stream);
int atomnas_reduce_flush(void* stream);

/* Fold jobs (ABI 9): dst[r][c] += src[r][c]; src[r][c] = 0 over a table of 2-D fp32 blocks, one launch per table slice.  The fused
 *   block of AtomNAS+ (models/mobilenet_base.py:236-254) keeps ONE contiguous expand / projection weight over all kernel-size groups,
 *   the kernels run on branch segments padded to whole 16-channel slabs: a layer's weight gradient is ONE atomnas_pw_gemm_tn into a
 *   padded scratch matrix, folded into the contiguous gradient tensor segment by segment here (the scratch is left zeroed).
 *   blk0 = first workgroup of the job, ascending, one workgroup per 256 elements; a launch covers jobs first .. first + njobs - 1 with
 *   blk_base = blk0 of job `first` and nblocks = the workgroups of the slice.  Blocks must not overlap. */
typedef struct atomnas_fold_job {
  float* src;
  float* dst;
  long src_ld, dst_ld;
  int rows, cols;
  unsigned blk0;
  int pad_;
} atomnas_fold_job;
int atomnas_fold_jobs(const void* jobs_dev, int first, int njobs, long blk_base, long nblocks, void* stream);

/* ---- dynamic shrink
 * alive masks |gamma| > thr (train.py:46-63, utils/prune.py:190-195): mode 0 current, 1 current|EMA, 2 EMA only.
 *   jobs_dev: struct { long off; int count; int out_off; };  outputs: mask bytes, ascending kept-channel indices, kept counts */
int atomnas_gamma_mask(const float* params, const float* ema, const void* jobs_dev, int njobs, float threshold, int mode,
                       unsigned char* mask, int* index, int* kept, void* stream);
/* channel repack in the reference's per-tensor protocol info['mask_hook'](new, old, mask), applied to weights, BN vectors,
 *   RMSprop square_avg / momentum_buffer (utils/rmsprop.py:134-165) and EMA shadows (utils/optim.py:134-153)
 *   (models/compress_utils.py:31-37): kept-channel index of a byte mask, then a gather along one dimension of an fp32 tensor
 *   viewed as [outer][dim][inner] with explicit element strides. */
int atomnas_mask_index(const unsigned char* mask, int count, int* index, int* kept, void* stream);
int atomnas_gather_dim(const float* src, float* dst, const int* index, long src_os, long src_ds, long dst_os, long dst_ds, int outer,
                       int n_kept, int inner, void* stream);
/* Job-list repack: every gather of one shrink (model weights, BatchNorm vectors, RMSprop state, EMA shadows;
 * models/compress_utils.py:31-37, utils/rmsprop.py:134-165, utils/optim.py:134-153) in ONE launch over a table in device memory.
 * A job is atomnas_gather_dim's arguments (index == NULL: identity, a strided copy); blk0 = the job's first workgroup, ascending, one
 * workgroup per 256 elements (outer * n_kept * inner); nblocks = the total.  Destinations must not overlap. */
typedef struct atomnas_gather_job {
  const float* src;
  float* dst;
  const int* index;
  long src_os, src_ds, dst_os, dst_ds;
  int outer, n_kept, inner;
  unsigned blk0;
} atomnas_gather_job;
int atomnas_gather_jobs(const void* jobs_dev, int njobs, long nblocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATOMNAS_HIP_H */

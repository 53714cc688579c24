// Round-4 experiment (not part of the product library): internal declarations shared by xdw_fused.hip and xdw_cw_bwd.hip
#pragma once
#include "../common.h"
namespace atomnas {
// dwconv_cw.hip: the depthwise backward with the expand output recomputed on chip (entry point atomnas_xdw_bwd in xdw.hip)
int xdw_cw_bwd(const void* gup, long gss, const void* yraw, long yrss, const float* c1, const float* c2, const float* c3, const void* xin,
               int ldx, int inp, const void* wexp, int ldwe, const float* sc, const float* sh, int relu, const float* w, int ldw, void* h,
               long hss, float* dw, float* stats, int stat_ld, int part_rows, float* dw_ws, int N, int H, int W, int C, int k,
               hipStream_t st);
int xdw_cw_bwd_supported(int N, int H, int W, int C, int k);
}  // namespace atomnas

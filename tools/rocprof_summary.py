"""Aggregates a rocprofv3 `--kernel-trace --output-format csv` trace per kernel family.

    python tools/rocprof_summary.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv [bench.log] > profiles/rNN_....txt

The template instances of one __global__ function (k_dwconv_bwd<bf16,7,1,7,16>, ...) are one family: that is the unit
bench.py's `roofline` object reports (one C-ABI entry point), so the average durations are directly comparable.
"""
import csv, re, sys, collections

FAMILIES = ["k_dwb_mm", "k_dwf_mm2", "k_dwf_mm", "k_gemm_nt_swg", "k_image_preprocess", "k_gather_jobs", "k_dwb_cw2", "k_dwb_cw", "k_dwf_cw", "k_gemm_nt_st", "k_gemm_nt_sw", "k_bnbwd_apply", "k_expand_bwd_s", "k_expand_bwd", "k_gemm_nt_small", "k_zero", "k_add_i64", "k_scale_by", "k_dwconv_bwd", "k_dwconv_fwd", "k_gemm_nt_ws", "k_gemm_nt", "k_gemm_tn3", "k_gemm_tn2", "k_gemm_tn", "k_reduce_batch", "k_gram_part", "k_gram_reduce", "k_xb_coeffs", "k_se_mlp", "k_se_pool", "k_se_wgrad", "k_se_bwd_apply", "k_se_scale", "k_act_bwd_stats",
            "k_bn_finalize_bwd", "k_bn_finalize_fwd", "k_bn_apply", "k_bn_act_pool", "k_pool_act_bwd", "k_bn_eval_coeffs",
            "k_reg_value", "k_reg_grad", "k_pack", "k_im2col_stem", "k_rmsprop_ema", "k_ema", "k_ce_smooth", "k_colsum",
            "k_gamma_mask", "k_mask_index", "k_gather_dim", "k_reduce_parts", "k_sum_partials", "k_se_squeeze", "k_se_mlp_fwd",
            "k_se_scale", "k_se_dgate", "k_se_mlp_bwd_img", "k_se_wgrad", "k_se_bwd_apply"]
ENTRY = {"k_dwb_mm": "atomnas_dwconv_bwd", "k_dwf_mm2": "atomnas_dwconv_fwd", "k_dwf_mm": "atomnas_dwconv_fwd", "k_gemm_nt_swg": "atomnas_pw_gemm_nt", "k_expand_bwd_s": "atomnas_expand_bwd", "k_gemm_nt_sw": "atomnas_pw_gemm_nt", "k_dwb_cw2": "atomnas_dwconv_bwd", "k_dwb_cw": "atomnas_dwconv_bwd", "k_dwf_cw": "atomnas_dwconv_fwd", "k_gemm_nt_st": "atomnas_pw_gemm_nt", "k_expand_bwd": "atomnas_expand_bwd", "k_gemm_nt_small": "atomnas_pw_gemm_nt", "k_gemm_nt_ws": "atomnas_pw_gemm_nt", "k_gemm_nt": "atomnas_pw_gemm_nt", "k_gemm_tn3": "atomnas_pw_gemm_tn", "k_gemm_tn2": "atomnas_pw_gemm_tn",
         "k_gemm_tn": "atomnas_pw_gemm_tn", "k_dwconv_bwd": "atomnas_dwconv_bwd", "k_dwconv_fwd": "atomnas_dwconv_fwd"}


def family(name):
    for f in FAMILIES:
        if re.search(r"(\b|\d)" + f + r"(\b|I|<)", name):
            return f
    return "other: " + name[:60]


def main():
    path = sys.argv[1]
    agg = collections.OrderedDict()
    with open(path) as fh:
        for row in csv.DictReader(fh):
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            a = agg.setdefault(family(row["Kernel_Name"]), [0, 0, 1 << 62, 0, 0, 0])
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
            a[4] = max(a[4], int(row["VGPR_Count"]) + int(row["Accum_VGPR_Count"])); a[5] = max(a[5], int(row["Scratch_Size"]))
    tot = sum(a[1] for a in agg.values())
    print("# rocprofv3 --kernel-trace, aggregated per kernel family by tools/rocprof_summary.py")
    print("# source: %s" % path)
    print("%-28s %7s %12s %11s %7s %10s %10s %6s %7s" % ("family", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us", "regs", "scratch"))
    for f, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %7d %12.3f %11.2f %6.2f%% %10.2f %10.2f %6d %7d" % (f, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / tot,
                                                                      a[2] / 1e3, a[3] / 1e3, a[4], a[5]))
    ent = collections.OrderedDict()
    for f, a in agg.items():
        e = ENTRY.get(f)
        if e:
            x = ent.setdefault(e, [0, 0]); x[0] += a[0]; x[1] += a[1]
    print("\n# per C-ABI entry point (what bench.py's roofline object names)")
    for e, x in ent.items():
        print("%-28s calls %6d  avg %9.2f us  total %10.3f ms" % (e, x[0], x[1] / x[0] / 1e3, x[1] / 1e6))
    if len(sys.argv) > 2:
        for line in open(sys.argv[2]):
            if line.startswith("{"):
                print("\n# bench.py line of the profiled run\n" + line.strip())


if __name__ == "__main__":
    main()

"""HBM traffic per launch of the hot kernels from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs).

    python tools/pmc_traffic.py <dir with *counter_collection.csv> > profiles/rNN_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-like
units of 1 KB per count as reported by rocprofv3; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane)
coalesced reads at 64 B, i.e. reports half of the bytes -> doubled here.  WRITE_SIZE is taken as reported.
Calibration in this repository: torch's fp32->bf16 conversion kernel of a 115.6 MB tensor reports FETCH_SIZE 56.5 MB and
WRITE_SIZE 56.5 MB for 57.8 MB written (gpurun_out/pmc_dwb, round 1).
"""
import csv, glob, json, re, sys, collections

FAMILIES = {"k_dwb_mm": "atomnas_dwconv_bwd", "k_dwf_mm2": "atomnas_dwconv_fwd", "k_dwf_mm": "atomnas_dwconv_fwd", "k_gemm_nt_swg": "atomnas_pw_gemm_nt", "k_expand_bwd_s": "atomnas_expand_bwd", "k_gemm_nt_sw": "atomnas_pw_gemm_nt", "k_dwb_cw2": "atomnas_dwconv_bwd", "k_dwb_cw": "atomnas_dwconv_bwd", "k_dwf_cw": "atomnas_dwconv_fwd",
            "k_gemm_nt_st": "atomnas_pw_gemm_nt",   # (its ST_PBWD instances run behind atomnas_project_bwd: a few launches of 78)
            "k_expand_bwd": "atomnas_expand_bwd", "k_gemm_nt_small": "atomnas_pw_gemm_nt",
            "k_dwconv_bwd": "atomnas_dwconv_bwd", "k_dwconv_fwd": "atomnas_dwconv_fwd", 
            "k_gemm_nt_ws": "atomnas_pw_gemm_nt", "k_gemm_nt": "atomnas_pw_gemm_nt", "k_gemm_tn3": "atomnas_pw_gemm_tn", "k_gemm_tn2": "atomnas_pw_gemm_tn",
            "k_gemm_tn": "atomnas_pw_gemm_tn"}


def family(name):
    for f, e in FAMILIES.items():
        if re.search(r"(\b|\d)" + f + r"(\b|I|<)", name):
            return e
    return None


root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        e = family(row.get("Kernel_Name", ""))
        if e is None:
            continue
        a = agg[e][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
out = {}
for e, d in agg.items():
    r = {}
    if "FETCH_SIZE" in d:
        n, v = d["FETCH_SIZE"]
        r["launches_sampled"] = n
        r["fetch_bytes_per_launch"] = 2.0 * v * 1024 / n     # gfx950 correction: x2
    if "WRITE_SIZE" in d:
        n, v = d["WRITE_SIZE"]
        r["write_bytes_per_launch"] = v * 1024 / n
    r["hbm_bytes_per_launch"] = r.get("fetch_bytes_per_launch", 0) + r.get("write_bytes_per_launch", 0)
    out[e] = r
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import build as _build  # noqa: E402
print(json.dumps({"lib_src_sha": _build.sources_digest(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 2 --warmup 1`, "
                            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte requests at 64 B)", "kernels": out}, indent=1))

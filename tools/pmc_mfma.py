"""MFMA counters per kernel family from a rocprofv3 --pmc pass (SQ_INSTS_VALU_MFMA_MOPS_BF16, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES).

    python tools/pmc_mfma.py <dir with *counter_collection.csv> > profiles/rNN_pmc_mfma.json

Counter values are summed over the dispatches of a family.  SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs (it counts the
cycles an MFMA occupies its SIMD's matrix pipe: 32 per 32x32x16 / 16x16x32 bf16 instruction, MI355X_MICROARCH.md), so
`mfma_util` = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch durations of the same pass x 2.4 GHz x 1024 SIMDs): the fraction of the chip's
matrix-pipe cycles the family used while it ran (durations from the counter pass itself, i.e. with the profiler attached).
MOPS are reported as counted (512-flop operations), `tflops` = MOPS x 512 / duration.
"""
import csv, glob, json, sys, collections
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import family, ENTRY

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        fam = family(row.get("Kernel_Name", ""))
        agg[fam][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row.get("Dispatch_Id"), fam)
        if key not in seen:
            seen.add(key); calls[fam] += 1
            agg[fam]["duration_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
out = {}
for fam, d in agg.items():
    if d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) <= 0:
        continue
    r = {"entry": ENTRY.get(fam), "dispatches": calls[fam]}
    r.update({k: v for k, v in d.items()})
    if d.get("duration_ns"):
        r["mfma_util"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["duration_ns"] * 1e-9 * 2.4e9 * 1024)
        r["tflops"] = d["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / (d["duration_ns"] * 1e-9) / 1e12
    out[fam] = r
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import build as _build  # noqa: E402
print(json.dumps({"lib_src_sha": _build.sources_digest(), "source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES over "
                            "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline` (its own pass)", "families": out}, indent=1))

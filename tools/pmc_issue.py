"""Instruction-issue counters per kernel family from the two rocprofv3 --pmc passes of tools/pmc_issue.sh.

    python tools/pmc_issue.py gpurun_out/pmc_issue

Per family (sorted by time): dispatches, time, instructions per wave (VALU / SALU / LDS / VMEM), the share of its wave-cycles a wave spent
issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES) and waiting on a counter (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES), the waves resident per SIMD
on average (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES, both in quad-cycles summed over the chip: an estimate), and instructions issued per SIMD and
cycle at 2.4 GHz (all classes / (duration x 1024 SIMDs x 2.4 GHz)).  A family near 0.2-0.25 instructions per SIMD-cycle with one or two
waves per SIMD is bound by what its waves issue (a wave issues at most one instruction every 4-5 cycles), not by bytes."""
import csv, glob, sys, collections
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import family

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ("A", "B"):
    for f in glob.glob(root + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            fam = family(row.get("Kernel_Name", ""))
            agg[fam][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row.get("Dispatch_Id"), fam)
            if key not in seen:
                seen.add(key)
                agg[fam]["n_" + sub] += 1
                agg[fam]["ns_" + sub] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
print("%-22s %5s %9s | %8s %7s %6s %6s | %6s %6s %6s %6s" % ("family", "calls", "ms", "valu/wv", "salu/wv", "lds/wv", "vmem/wv", "issue", "wait", "wv/simd", "ipc"))
for fam, d in sorted(agg.items(), key=lambda kv: -kv[1].get("ns_A", 0)):
    if d.get("ns_A", 0) < 2e5 or not d.get("SQ_WAVES"):
        continue
    w = d["SQ_WAVES"]
    ins = d.get("SQ_INSTS_VALU", 0) + d.get("SQ_INSTS_SALU", 0) + d.get("SQ_INSTS_LDS", 0) + d.get("SQ_INSTS_VMEM", 0)
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-22s %5d %9.3f | %8.0f %7.0f %6.0f %6.0f | %6.2f %6.2f %6.2f %6.3f" % (
        fam[:22], d["n_A"], d["ns_A"] * 1e-6, d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_LDS", 0) / w,
        d.get("SQ_INSTS_VMEM", 0) / w, d.get("SQ_ACTIVE_INST_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc,
        wc / (d.get("SQ_BUSY_CYCLES", 0) or 1) , ins / (d["ns_A"] * 1e-9 * 1024 * 2.4e9)))

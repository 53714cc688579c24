"""Idle time inside the replayed training step: kernel trace (rocprofv3 --kernel-trace --output-format csv of bench.py) -> for the last
replayed steps: wall time from the first kernel's start to the last kernel's end, the sum of kernel durations, the idle gaps between
consecutive kernels (one queue), and the gaps grouped by the kernel that FOLLOWS them.

    python tools/graph_gaps.py gpurun_out/prof_dir [nsteps]
Trace a run WITHOUT the eager passes of the roofline / cpu legs (bench.py --no-roofline --no-cpu-baseline): every step of such a run
after the warm-up is a graph replay.
"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"^void ", "", re.sub(r"[<(].*", "", r["Kernel_Name"])).replace("atomnas::", "")
loss = [i for i, r in enumerate(rows) if "k_ce_smooth" in r["Kernel_Name"]]
# a step = the kernels from one loss kernel to the next (same position inside consecutive steps)
for s in range(len(loss) - nsteps - 1, len(loss) - 1):
    seg = rows[loss[s]:loss[s + 1]]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[loss[s + 1]]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = collections.defaultdict(lambda: [0, 0.0])
    tot_gap = 0.0
    for a, b in zip(seg, seg[1:] + [rows[loss[s + 1]]]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 0:
            e = gaps[name(b)]
            e[0] += 1
            e[1] += g
            tot_gap += g
    print("step %d: %d kernels, wall %.3f ms, kernel time %.3f ms, idle %.3f ms (%.1f %%), mean gap %.2f us" % (
        s, len(seg), (t1 - t0) / 1e6, busy / 1e6, tot_gap / 1e6, 100.0 * tot_gap / (t1 - t0), tot_gap / 1e3 / len(seg)))
for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   idle before %-28s n=%4d  %8.1f us  (%.2f us each)" % (k, c, g / 1e3, g / 1e3 / c))

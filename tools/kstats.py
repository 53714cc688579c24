"""Per-kernel summary of a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py.

    python tools/kstats.py gpurun_out/prof_dir [substring] [--launches]
Kernel time per step (divides by the number of steps the process ran: timed + warm-up + capture), calls per step; with a
substring and --launches, the per-launch durations and grids of the last step.
"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
rows = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
names = collections.Counter(r["Kernel_Name"] for r in rows)
# forward-backward passes the process ran (timed + warm-up + the eager passes before the capture) = launches of the loss kernel
steps = min((c for n, c in names.items() if "k_ce_smooth" in n), default=1)
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    a = agg[n]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[1] += 1
tot = sum(a[0] for a in agg.values())
print("steps %d, kernel time %.3f ms/step" % (steps, tot / 1e6 / steps))
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:(40 if sub is None else 10000)]:
    if sub is None or sub in n:
        print("%8.3f ms/step %6.1f calls/step %7.1f us  %s" % (t / 1e6 / steps, c / steps, t / 1e3 / c, n[:90]))
if sub:
    print("sum over '%s': %.3f ms/step" % (sub, sum(t for n, (t, c) in agg.items() if sub in n) / 1e6 / steps))
if "--launches" in sys.argv and sub:
    by = collections.defaultdict(list)
    for r in rows:
        if sub in r["Kernel_Name"]:
            by[re.sub(r"\(.*", "", r["Kernel_Name"])[:70]].append(r)
    for k, v in by.items():
        v.sort(key=lambda r: int(r["Start_Timestamp"]))
        last = v[-(len(v) // steps):]
        print(k)
        print("   us  ", " ".join("%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in last))
        print("   grid", " ".join("%dx%sx%s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"]) for r in last))

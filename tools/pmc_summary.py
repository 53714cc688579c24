"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (mean over dispatches)."""
import csv, glob, sys, collections, re
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if want and want not in k: continue
        k = re.sub(r"\(.*", "", k)[-60:]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))

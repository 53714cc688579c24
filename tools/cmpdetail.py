"""Per-shape comparison of tools/bringup.py DETAIL logs:  python tools/cmpdetail.py base.log other1.log other2.log ... [filter]"""
import re, sys
def parse(f):
    d = {}
    for l in open(f):
        m = re.match(r"\s+(\w+)\s+(.*?)\s+n=\s*(\d+)\s+([\d.]+) ms", l)
        if m and m.group(1) in ('dwconv_fwd', 'dwconv_bwd', 'pw_gemm_nt', 'pw_gemm_tn'):
            k = (m.group(1), m.group(2).strip())
            d[k] = d.get(k, 0) + float(m.group(4))
    return d
files = [a for a in sys.argv[1:] if a.endswith(".log")]
flt = [a for a in sys.argv[1:] if not a.endswith(".log")]
ds = [parse(f) for f in files]
tot = [dict() for _ in ds]
for k in ds[0]:
    if flt and not any(x in k[0] for x in flt): continue
    print("%-11s %-40s" % k + " ".join("%7.3f" % d.get(k, 0) for d in ds))
    for i, d in enumerate(ds): tot[i][k[0]] = tot[i].get(k[0], 0) + d.get(k, 0)
for n in tot[0]: print("TOTAL %-46s" % n + " ".join("%7.3f" % t.get(n, 0) for t in tot))

"""Prints VGPR/AGPR/scratch/occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", src, "-o", "/tmp/_ru.o",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-z/SIMD]+\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    n = r["name"]
    if flt and flt not in n: continue
    n = re.sub(r"^_ZN7atomnas\d+", "", n)
    n = re.sub(r"EEvNS_7Operand.*|EEvPKT_.*", "", n)
    print("%-44s vgpr %3d agpr %3d scratch %4d occ %2d lds %6d sgpr %3d" % (n[:44], r.get("VGPRs", -1), r.get("AGPRs", -1),
          r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1), r.get("TotalSGPRs", -1)))

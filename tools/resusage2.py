"""Resource usage per kernel from a hipcc -Rpass-analysis=kernel-resource-usage log (stderr of the compile):
    python tools/resusage2.py build.log [filter]"""
import re, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark: \S+\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    n = r["name"]
    if flt and flt not in n: continue
    n = re.sub(r"^_ZN7atomnas\d+", "", n); n = re.sub(r"EEv.*", "", n)
    print("%-36s vgpr %3s sgpr %3s scratch %5s occ %2s spillV %4s spillS %4s" % (n[:36], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"),
          r.get("Occupancy"), r.get("VGPRs Spill"), r.get("SGPRs Spill")))

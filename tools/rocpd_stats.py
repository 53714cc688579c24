"""Summarises a rocprofv3 rocpd database (kernel trace) into a per-kernel table: calls, total, average, share."""
import re
import sqlite3
import sys


import shutil
import subprocess

_CXXFILT = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
_cache = {}


def short(name):
    if name in _cache:
        return _cache[name]
    full = name
    if name.startswith("_Z"):
        try:
            full = subprocess.run([_CXXFILT, name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    full = re.sub(r"^void ", "", full)
    m = re.match(r"(?:atomnas::)?(k_[a-z_0-9]+)(<.*?>)?\(", full)
    if m:
        out = m.group(1) + (m.group(2) or "").replace("__bf16", "bf16").replace(" ", "")
    else:
        out = re.sub(r"\(.*$", "", full).split("::")[-1][:90]
    _cache[name] = out
    return out


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += (e - s)
    tot = sum(v[1] for v in agg.values())
    lines = ["%-70s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%")]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-70s %8d %12.1f %10.2f %6.2f" % (k[:70], v[0], v[1] / 1e3, v[1] / 1e3 / v[0], 100.0 * v[1] / tot))
    lines.append("TOTAL kernel time %.1f us over %d dispatches" % (tot / 1e3, len(rows)))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

"""ISA check for the asynchronous inline-asm loads of the depthwise kernels.

cw_row_issue / the ds_read_b64 blocks of dwconv.hip issue `s_load_dwordx2` and `ds_read_b64` from inline asm whose results the
compiler believes to be valid at once; the code guarantees validity with a manual `s_waitcnt lgkmcnt(0)` before the first use.  A
compiler that copied, spilled or overwrote one of those registers between issue and wait would read stale data silently.  This script
scans the gfx950 assembly of a translation unit and reports every instruction that touches a register with such a load in flight:

    python tools/check_asm_waits.py atomnas_amd/csrc/dwconv_cw.hip      (compiles with the build's flags, -S)  -> exit 1 on a finding

Linear scan per function: registers written by an asm load are pending until an `s_waitcnt` with lgkmcnt(0) (or a bare
`s_waitcnt lgkmcnt(0)`), and no instruction outside ASMSTART/ASMEND may name a pending register.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([vs])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        k = m.group(1)
        if m.group(2) is not None:
            out.add((k, int(m.group(2))))
        else:
            out.update((k, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def assemble(src):
    sys.path.insert(0, ROOT)
    from atomnas_amd import build
    out = os.path.join(build.OBJ_DIR, os.path.basename(src)[:-4] + ".s")
    stamp = out + ".sha1"
    dig = build._digest(src)
    if not (os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig):
        os.makedirs(build.OBJ_DIR, exist_ok=True)
        subprocess.run([build._hipcc()] + build.FLAGS + ["--cuda-device-only", "-S", src, "-o", out], check=True, capture_output=True)
        open(stamp, "w").write(dig)
    return out


def check(path):
    findings, func, in_asm, pending, nloads = [], None, False, set(), 0
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if re.match(r"^_Z\w+:", t):
            func, pending = t.split(":")[0], set()
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        rest = rest.split(";")[0]
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in rest:
                pending = set()
            continue
        if in_asm:
            if op in ("ds_read_b64", "s_load_dwordx2", "s_load_dwordx4", "ds_read_b128", "ds_read_b32"):
                pending |= regs(rest.split(",")[0])
                nloads += 1
            continue
        hit = regs(rest) & pending
        if hit:
            findings.append("%s:%d %s: `%s` touches %s while its asm load is in flight" % (os.path.basename(path), ln, func, t, sorted(hit)[:4]))
    return findings, nloads


if __name__ == "__main__":
    bad = []
    for src in sys.argv[1:]:
        f, n = check(assemble(os.path.abspath(src)))
        print("%s: %d asynchronous asm loads checked, %d findings" % (src, n, len(f)))
        bad += f
    for b in bad[:20]:
        print(b)
    sys.exit(1 if bad else 0)

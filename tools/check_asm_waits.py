"""ISA check for the asynchronous inline-asm loads of the depthwise kernels.

cw_row_issue / the ds_read_b64 blocks of dwconv.hip issue `s_load_dwordx2` and `ds_read_b64` from inline asm whose results the
compiler believes to be valid at once; the code guarantees validity with a manual `s_waitcnt lgkmcnt(0)` before the first use.  A
compiler that copied, spilled or overwrote one of those registers between issue and wait would read stale data silently.  This script
scans the gfx950 assembly of a translation unit and reports every instruction that touches a register with such a load in flight:

    python tools/check_asm_waits.py atomnas_amd/csrc/dwconv_cw.hip      (compiles with the build's flags, -S)  -> exit 1 on a finding

Per function a forward data-flow over the basic blocks: registers written by an asm load are pending until an `s_waitcnt` with
lgkmcnt(0), and no instruction outside ASMSTART/ASMEND may name a pending register on any path.
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


ASM_LOADS = ("ds_read_b64", "s_load_dwordx2", "s_load_dwordx4", "ds_read_b128", "ds_read_b32", "ds_read_b64_tr_b16")


def functions(path):
    """[(name, [(line number, text, inside inline asm)])] per kernel of the assembly file"""
    out, cur, in_asm = [], None, False
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if re.match(r"^_Z\w+:", t):
            cur = (t.split(":")[0], [])
            out.append(cur)
            in_asm = False
            continue
        if cur is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        if not t or t[0] == ";" or (t[0] == "." and not t.endswith(":")):
            continue
        cur[1].append((ln, t.split(";")[0].strip() if not t.endswith(":") else t, in_asm))
    return out


def check(path):
    """Forward data-flow over the basic blocks of every kernel: `pending` = registers written by an inline-asm load that no
    `s_waitcnt ... lgkmcnt(0)` has covered yet; at a block entry the union over its predecessors (the assembler lays blocks out of
    line: a linear scan would see the consumers of a conditional read before its wait).  A finding = an instruction outside inline
    asm that names a pending register."""
    findings, nloads = [], 0
    for func, ins in functions(path):
        # basic blocks
        blocks, label_of, cur = [], {}, []
        for item in ins:
            ln, t, ia = item
            if t.endswith(":"):
                if cur:
                    blocks.append(cur)
                cur = []
                label_of[t[:-1]] = len(blocks)
                continue
            cur.append(item)
            op = t.split(" ")[0]
            if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                blocks.append(cur)
                cur = []
        if cur:
            blocks.append(cur)
        # a label may point at an index one past the blocks emitted so far: it names the NEXT block
        succ = [[] for _ in blocks]
        for i, blk in enumerate(blocks):
            if not blk:
                if i + 1 < len(blocks):
                    succ[i].append(i + 1)
                continue
            t = blk[-1][1]
            op = t.split(" ")[0]
            tgt = t.split(" ")[-1] if (op.startswith("s_cbranch") or op == "s_branch") else None
            if tgt is not None and tgt in label_of and label_of[tgt] < len(blocks):
                succ[i].append(label_of[tgt])
            if op not in ("s_branch", "s_endpgm", "s_setpc_b64") and i + 1 < len(blocks):
                succ[i].append(i + 1)
        entry = [set() for _ in blocks]
        work = list(range(len(blocks)))
        reported = set()
        while work:
            i = work.pop()
            pending = set(entry[i])
            for ln, t, ia in blocks[i]:
                op, _, rest = t.partition(" ")
                if op == "s_waitcnt":
                    if "lgkmcnt(0)" in rest:
                        pending = set()
                    continue
                if ia:
                    if op in ASM_LOADS:
                        pending |= regs(rest.split(",")[0])
                    continue
                hit = regs(rest) & pending
                if hit and ln not in reported:
                    reported.add(ln)
                    findings.append("%s:%d %s: `%s` touches %s while its asm load is in flight" % (os.path.basename(path), ln, func, t, sorted(hit)[:4]))
            for j in succ[i]:
                if not pending <= entry[j]:
                    entry[j] |= pending
                    work.append(j)
        nloads += sum(1 for _, t, ia in ins if ia and t.split(" ")[0] in ASM_LOADS)
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

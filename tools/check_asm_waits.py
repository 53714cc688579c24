"""ISA check for the asynchronous inline-asm loads of the depthwise kernels.

cw_row_issue / the ds_read_b64 blocks of dwconv.hip issue `s_load_dwordx2` and `ds_read_b64` from inline asm whose results the
compiler believes to be valid at once; the code guarantees validity with a manual `s_waitcnt lgkmcnt(0)` before the first use.  A
compiler that copied, spilled or overwrote one of those registers between issue and wait would read stale data silently.  This script
scans the gfx950 assembly of a translation unit and reports every instruction that touches a register with such a load in flight:

    python tools/check_asm_waits.py atomnas_amd/csrc/dwconv_cw.hip      (compiles with the build's flags, -S)  -> exit 1 on a finding

Per function a forward data-flow over the basic blocks: registers written by an asm load are pending until an `s_waitcnt` with
lgkmcnt(0), and no instruction outside ASMSTART/ASMEND may name a pending register on any path.

Round 5: the LDS-DMA rings (k_expand_bwd_s, k_gemm_nt_sw, k_gemm_tn3 of pwconv.hip, k_gram_part of xbwd.hip) wait for their copies with
a COUNTED `s_waitcnt vmcnt(N)`.  That is right only if (check_rings)
  * every pass of the loop that holds the wait issues the same number CPS of `global_load_lds_dwordx4` copies on every path,
  * N is a multiple of CPS and exactly N + CPS copies were issued on every path from the kernel entry to that loop (the ring is
    N / CPS + 1 stages deep when the first wait runs: the oldest stage is what the wait completes),
  * the compiler put no vector-memory LOAD of its own into that loop (its wait would ignore the copies and, memory operations
    retiring in order, drain the ring; stores only make the counted wait conservative).
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


def _cfg(ins):
    """basic blocks of one function: (blocks, successor lists)"""
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
    return blocks, succ


COPY = "global_load_lds_dwordx4"
VMEM_LOAD = ("global_load_", "buffer_load_", "flat_load_", "scratch_load_")


def check_rings(path):
    """-> (findings, [description of every ring found])"""
    findings, rings = [], []
    for func, ins in functions(path):
        if not any(ia and t.startswith(COPY) for _, t, ia in ins):
            continue
        blocks, succ = _cfg(ins)
        n = len(blocks)
        # back edges by depth-first search from the entry (an edge to a block on the search stack)
        back, state, stack = set(), [0] * n, [(0, 0)]
        state[0] = 1
        while stack:
            i, k = stack[-1]
            if k < len(succ[i]):
                stack[-1] = (i, k + 1)
                j = succ[i][k]
                if state[j] == 1:
                    back.add((i, j))
                elif state[j] == 0:
                    state[j] = 1
                    stack.append((j, 0))
            else:
                state[i] = 2
                stack.pop()
        pred = [[] for _ in range(n)]
        for i in range(n):
            for j in succ[i]:
                pred[j].append(i)

        def natural_loop(latch, header):
            body, work = {header, latch}, [latch]
            while work:
                b = work.pop()
                if b == header:
                    continue
                for q in pred[b]:
                    if q not in body:
                        body.add(q)
                        work.append(q)
            return body
        loops = {}
        for (latch, header) in back:
            loops.setdefault(header, set()).update(natural_loop(latch, header))
        copies = [sum(1 for _, t, ia in blk if ia and t.startswith(COPY)) for blk in blocks]
        # forward DAG (back edges removed): min / max copies on the paths from `src` into (not including) every block
        order, seen = [], [False] * n

        def topo(i):
            seen[i] = True
            for j in succ[i]:
                if (i, j) not in back and not seen[j]:
                    topo(j)
            order.append(i)
        sys.setrecursionlimit(max(10000, 4 * n))
        topo(0)
        order.reverse()

        def path_counts(src, allowed=None):
            lo, hi = {src: 0}, {src: 0}
            for i in order:
                if i not in lo or (allowed is not None and i not in allowed):
                    continue
                for j in succ[i]:
                    if (i, j) in back or (allowed is not None and j not in allowed):
                        continue
                    a, b = lo[i] + copies[i], hi[i] + copies[i]
                    lo[j] = min(lo.get(j, a), a)
                    hi[j] = max(hi.get(j, b), b)
            return lo, hi
        for bi, blk in enumerate(blocks):
            for ln, t, ia in blk:
                m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
                if not (ia and m and int(m.group(1)) > 0):
                    continue
                N = int(m.group(1))
                cands = [(len(body), h) for h, body in loops.items() if bi in body]
                if not cands:
                    findings.append("%s:%d %s: counted wait vmcnt(%d) outside any loop" % (os.path.basename(path), ln, func, N))
                    continue
                _, h = min(cands)
                body = loops[h]
                if any(h2 != h and h2 in body and any(copies[b] for b in loops[h2]) for h2 in loops):
                    findings.append("%s:%d %s: ring copies inside a loop nested in the wait's loop" % (os.path.basename(path), ln, func))
                    continue
                lo, hi = path_counts(h, body)
                latches = [i for (i, j) in back if j == h]
                per = {(lo[i] + copies[i], hi[i] + copies[i]) for i in latches if i in lo}
                if len(per) != 1 or next(iter(per))[0] != next(iter(per))[1]:
                    findings.append("%s:%d %s: copies per pass of the ring loop differ by path: %s" % (os.path.basename(path), ln, func, sorted(per)))
                    continue
                cps = next(iter(per))[0]
                plo, phi = path_counts(0)
                pro = (plo.get(h), phi.get(h))
                own = [t2 for b in body for _, t2, ia2 in blocks[b] if not ia2 and t2.split(" ")[0].startswith(VMEM_LOAD)]
                desc = "%s: wait vmcnt(%d), %d copies per pass, %s copies before the loop" % (func[:60], N, cps, pro[0] if pro[0] == pro[1] else pro)
                rings.append(desc)
                if cps == 0 or N % cps != 0 or pro[0] != pro[1] or pro[0] != N + cps:
                    findings.append("%s:%d ring miscount -- %s (expected N %% CPS == 0 and N + CPS copies before the loop)" % (os.path.basename(path), ln, desc))
                if own:
                    findings.append("%s:%d %s: compiler-emitted vector-memory load inside the ring loop: `%s`" % (os.path.basename(path), ln, func, own[0]))
    return findings, rings


if __name__ == "__main__":
    bad = []
    for src in sys.argv[1:]:
        asm = src if src.endswith(".s") else assemble(os.path.abspath(src))
        f, n = check(asm)
        print("%s: %d asynchronous asm loads checked, %d findings" % (src, n, len(f)))
        bad += f
        f, rings = check_rings(asm)
        print("%s: %d LDS-DMA rings with counted waits checked, %d findings" % (src, len(rings), len(f)))
        for r in rings:
            print("    " + r)
        bad += f
    for b in bad[:20]:
        print(b)
    sys.exit(1 if bad else 0)

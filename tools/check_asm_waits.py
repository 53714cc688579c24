"""ISA check for the asynchronous inline-asm loads of the depthwise kernels.

cw_row_issue / the ds_read_b64 blocks of dwconv.hip issue `s_load_dwordx2` and `ds_read_b64` from inline asm whose results the
compiler believes to be valid at once; the code guarantees validity with a manual `s_waitcnt lgkmcnt(0)` before the first use.  A
compiler that copied, spilled or overwrote one of those registers between issue and wait would read stale data silently.  This script
scans the gfx950 assembly of a translation unit and reports every instruction that touches a register with such a load in flight:

    python tools/check_asm_waits.py atomnas_amd/csrc/dwconv_cw.hip      (compiles with the build's flags, -S)  -> exit 1 on a finding

Per function a forward data-flow over the basic blocks: registers written by an asm load are pending until an `s_waitcnt` with
lgkmcnt(0) -- or, round 6, a COUNTED lgkmcnt(N) that provably covers them (LDS operations return in order: everything but the N
operations issued last, when no scalar-memory load is in flight) --, and no instruction outside ASMSTART/ASMEND may name a pending
register on any path.

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
    return build.assemble(src)


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
        m = re.match(r"^(\.L\w+):", t)      # a label, possibly followed by a comment ("; =>This Inner Loop Header: Depth=1")
        if m:
            cur[1].append((ln, m.group(1) + ":", in_asm))
            continue
        if not t or t[0] == ";" or t[0] == ".":
            continue
        cur[1].append((ln, t.split(";")[0].strip(), in_asm))
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
        smem_entry = [False for _ in blocks]   # a scalar-memory load may be in flight at the block's entry
        work = list(range(len(blocks)))
        reported = set()
        while work:
            i = work.pop()
            pending = set(entry[i])
            # Counted waits (round 6, k_gemm_nt_swg): LDS operations of a wave return in order, so `s_waitcnt lgkmcnt(N)` completes every
            # LDS operation but the N issued last.  `events` = the lgkm-class operations of THIS block in issue order (asm loads with their
            # destination registers, compiler-issued ds_* / scalar-memory operations with none); a counted wait retires the block's own
            # events but the last N -- and everything pending from the predecessors, which is older -- when N does not reach past the
            # block's own events and no scalar-memory load (they may return out of order) is among what is in flight.
            events, smem_in_flight = [], bool(smem_entry[i])
            for ln, t, ia in blocks[i]:
                op, _, rest = t.partition(" ")
                if op == "s_waitcnt":
                    m = re.search(r"lgkmcnt\((\d+)\)", rest)
                    if m is not None:
                        n = int(m.group(1))
                        if n == 0:
                            pending, events, smem_in_flight = set(), [], False
                        elif not smem_in_flight and n <= len(events):
                            done = events[:len(events) - n]
                            events = events[len(events) - n:]
                            still = set().union(*[e for e in events]) if events else set()
                            pending = still   # the predecessors' loads and this block's older ones are complete
                            del done
                    continue
                is_lgkm = op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_scratch_load")
                if is_lgkm and not op.startswith("ds_"):
                    smem_in_flight = True
                if ia:
                    if op in ASM_LOADS:
                        r = regs(rest.split(",")[0])
                        pending |= r
                        events.append(r)
                    elif is_lgkm:
                        events.append(set())
                    continue
                if is_lgkm:
                    events.append(set())
                hit = regs(rest) & pending
                if hit and ln not in reported:
                    reported.add(ln)
                    findings.append("%s:%d %s: `%s` touches %s while its asm load is in flight" % (os.path.basename(path), ln, func, t, sorted(hit)[:4]))
            for j in succ[i]:
                if not pending <= entry[j] or (smem_in_flight and not smem_entry[j]):
                    entry[j] |= pending
                    smem_entry[j] = smem_entry[j] or smem_in_flight
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


MARK = "; atomnas_ring_stage_end"


def check_rings(path):
    """Counted waits of the LDS-DMA rings, as a queue model.  The kernels mark the end of every stage they issue
    (ATOMNAS_RING_STAGE_END, an assembler comment inside inline asm); stages are consumed first in first out, one per counted
    `s_waitcnt vmcnt(N)`.  Vector-memory operations retire in order, so the wait leaves the stage complete iff at least N copies were
    issued AFTER that stage's end marker.  State = for every stage not yet waited for, the copies issued since its marker; forward
    data-flow of the set of states per basic block (loops make it periodic).  A finding: a wait with fewer than N copies behind its
    stage (returns early: stale LDS data), a wait with no stage outstanding, or a compiler-emitted vector-memory LOAD inside a loop
    that holds a counted wait (its own wait would ignore the copies and drain the ring).  More than N copies behind the stage only
    makes the wait conservative (k_expand_bwd_s: the x / residual rows ride in the queue); reported in the description.
    -> (findings, [description of every ring kernel])"""
    findings, rings = [], []
    for func, raw in functions_with_marks(path):
        if not any(ia and t.startswith(COPY) for _, t, ia in raw):
            continue
        blocks, succ = _cfg(raw)
        n = len(blocks)
        base = os.path.basename(path)

        def events(blk):
            out = []
            for ln, t, ia in blk:
                if ia and t.startswith(COPY):
                    out.append(("c", ln, 0))
                elif ia and t == MARK:
                    out.append(("m", ln, 0))
                else:
                    m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
                    if ia and m and int(m.group(1)) > 0:
                        out.append(("w", ln, int(m.group(1))))
            return out
        ev = [events(b) for b in blocks]
        if not any(e[0] == "w" for b in ev for e in b):
            continue
        if not any(e[0] == "m" for b in ev for e in b):
            findings.append("%s %s: counted vmcnt waits but no ATOMNAS_RING_STAGE_END markers" % (base, func))
            continue
        val = [set() for _ in range(n)]
        val[0] = {()}
        work, noted, slack, slack_hi, sites, depth = [0], set(), {}, {}, set(), 0
        while work:
            i = work.pop()
            outs = set()
            for st in val[i]:
                st = list(st)
                for kind, ln, N in ev[i]:
                    if kind == "c":
                        st = [v + 1 for v in st]
                    elif kind == "m":
                        st.append(0)
                        depth = max(depth, len(st))
                    else:
                        sites.add(ln)
                        if not st:
                            if ln not in noted:
                                noted.add(ln)
                                findings.append("%s:%d %s: wait vmcnt(%d) with no stage outstanding" % (base, ln, func, N))
                            continue
                        behind = st.pop(0)
                        slack[ln] = min(slack.get(ln, behind - N), behind - N)
                        slack_hi[ln] = max(slack_hi.get(ln, behind - N), behind - N)
                        if behind < N and ln not in noted:
                            noted.add(ln)
                            findings.append("%s:%d %s: wait vmcnt(%d) but only %d copies were issued after the end of its stage: the wait "
                                            "can return before the stage has landed" % (base, ln, func, N, behind))
                outs.add(tuple(st))
            for j in succ[i]:
                if not outs <= val[j]:
                    val[j] |= outs
                    if len(val[j]) > 64 or any(len(t) > 16 for t in val[j]):
                        if ("u", j) not in noted:
                            noted.add(("u", j))
                            findings.append("%s %s: stages issued and counted waits are not balanced over a loop (block %d)" % (base, func, j))
                        continue
                    work.append(j)
        # compiler-emitted vector-memory loads inside a loop that holds a counted wait
        state, stack, back = [0] * n, [(0, 0)], set()
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
        for (latch, header) in back:
            body, wk = {header, latch}, [latch]
            while wk:
                q = wk.pop()
                if q == header:
                    continue
                for r in pred[q]:
                    if r not in body:
                        body.add(r)
                        wk.append(r)
            if not any(e[0] == "w" for q in body for e in ev[q]):
                continue
            own = [(ln, t2) for q in body for ln, t2, ia2 in blocks[q] if not ia2 and t2.split(" ")[0].startswith(VMEM_LOAD)]
            if own and ("l", own[0][0]) not in noted:
                noted.add(("l", own[0][0]))
                findings.append("%s:%d %s: compiler-emitted vector-memory load inside a ring loop: `%s`" % (base, own[0][0], func, own[0][1]))
        ws = sorted({e[2] for b in ev for e in b if e[0] == "w"})
        rings.append("%s: waits %s at %d sites, up to %d stages in flight, copies behind the awaited stage beyond N: %s" % (
            func[:64], ["vmcnt(%d)" % w for w in ws], len(sites), depth,
            "0 (exact)" if slack and max(slack_hi.values()) == 0 and min(slack.values()) == 0 else "%d..%d" % (min(slack.values()), max(slack_hi.values())) if slack else "-"))
    return findings, rings


def functions_with_marks(path):
    """functions(), keeping the stage-end marker comments of inline asm as instructions"""
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
        if in_asm and t == MARK:
            cur[1].append((ln, MARK, True))
            continue
        m = re.match(r"^(\.L\w+):", t)
        if m:
            cur[1].append((ln, m.group(1) + ":", in_asm))
            continue
        if not t or t[0] == ";" or t[0] == ".":
            continue
        cur[1].append((ln, t.split(";")[0].strip(), in_asm))
    return out


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

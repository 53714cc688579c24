"""Instruction counts of the outermost loops of every kernel in an assembly listing (development aid): python tools/loopstat.py file.s [name-filter]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and flt in l]


def isins(l):
    l = l.strip()
    return l and not l.startswith(('.', ';', '_')) and not l.endswith(':') and not l.startswith('s_nop')


for st in starts:
    name = lines[st].split(':')[0]
    end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
    body = lines[st:end]
    out = []
    for h, l in enumerate(body):
        if 'Loop Header: Depth=1' in l and l.startswith('.LBB'):
            label = re.match(r'\.L(BB\d+_\d+)', l).group(1)
            idx = [i for i, b in enumerate(body) if ('Header=%s ' % label) in b] or [h]
            j = max(idx) + 1
            while j < len(body) and not body[j].startswith('.LBB'):
                j += 1
            loop = [b for b in body[h:j] if isins(b)]
            if len(loop) > 100:
                out.append("loop %d instr (valu %d, salu %d, mfma %d, ds_read %d, ds_write %d, global %d)" % (
                    len(loop), sum(b.strip().startswith('v_') and 'mfma' not in b for b in loop), sum(b.strip().startswith('s_') for b in loop),
                    sum('mfma' in b for b in loop), sum(b.strip().startswith('ds_read') for b in loop),
                    sum(b.strip().startswith('ds_write') for b in loop), sum(b.strip().startswith(('global_', 'buffer_')) for b in loop)))
    print(name[:60], 'total', sum(isins(b) for b in body), '|', '; '.join(out))

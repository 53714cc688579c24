"""Instruction histogram of the loops of a kernel in an assembly listing (development aid).
    python tools/loophist.py file.s kernel-name-substring [min-instructions]
Every loop (any depth) of every matching kernel: instruction count and the most frequent opcodes.  Loop extent = header label to the last
backward branch to it (rotated loops included as the compiler laid them out)."""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and flt in l]
for st in starts:
    end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
    body = lines[st:end]
    print(lines[st].split(':')[0][:150])
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^\.L(BB\d+_\d+):', l)] if m}
    for lab, h in labels.items():
        if h + 1 >= len(body) or 'Loop Header' not in body[h] + body[h + 1]:
            continue
        back = [i for i, b in enumerate(body) if i > h and re.search(r's_c?branch\w* \.L%s\b' % lab, b)]
        if not back:
            continue
        loop = [b.strip() for b in body[h:max(back) + 1] if b.strip() and b.strip()[0] not in '.;' and not b.strip().endswith(':')]
        if len(loop) < minins:
            continue
        c = collections.Counter(b.split()[0] for b in loop)
        depth = re.search(r'Depth=(\d)', body[h] + body[h + 1])
        print("  loop %s depth %s: %d instr: %s" % (lab, depth.group(1) if depth else '?', len(loop),
                                                   ' '.join('%s:%d' % kv for kv in c.most_common(22))))

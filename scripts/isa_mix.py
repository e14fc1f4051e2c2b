#!/usr/bin/env python3
"""Instruction mix of the kernels in a `hipcc -S --cuda-device-only` listing whose mangled name contains every word
given: floating-point VALU, other VALU, LDS, global memory, scalar, waits, barriers (static counts, straight-line
kernels: every instruction runs once per thread unless it sits in a loop).  With --loops as the first argument the
counts are also given per loop (the compiler's "in Loop: Header=..." block comments), which is what matters for the
particle kernels.
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only pmx_colfft.hip -o /tmp/colfft.s
    python scripts/isa_mix.py /tmp/colfft.s colfft_round_kernelIfLi9ELb1"""
import collections
import re
import sys

LOOPS = len(sys.argv) > 1 and sys.argv[1] == '--loops'
if LOOPS:
    del sys.argv[1]
lines = open(sys.argv[1]).read().splitlines()
words = sys.argv[2:]


def classify(line):
    m = re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)(\w+)', line)
    if not m:
        return None
    op = m.group(0)
    if op.startswith('v_'):
        return 'valu_fp' if re.search(r'_f(32|64)', op) and 'cvt' not in op else 'valu_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'scalar'
    return 'vmem'


starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for n, (i, name) in enumerate(starts):
    if not all(w in name for w in words):
        continue
    j = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    c = collections.Counter()
    for line in lines[i:j]:
        line = line.strip()
        m = re.match(r'^(v_|s_|ds_|global_|buffer_|flat_)(\w+)', line)
        if m:
            op = m.group(0)
            if op.startswith('v_'):
                c['valu_fp' if re.search(r'_f(32|64)', op) and 'cvt' not in op else 'valu_other'] += 1
            elif op.startswith('ds_'):
                c['lds'] += 1
            elif op.startswith('s_waitcnt'):
                c['waitcnt'] += 1
            elif op.startswith('s_barrier'):
                c['barrier'] += 1
            elif op.startswith('s_'):
                c['scalar'] += 1
            else:
                c['vmem'] += 1
        if line.startswith('s_endpgm'):
            break
    print(name[:100], dict(c))
    if LOOPS:
        loops, cur = collections.OrderedDict(), None
        for line in lines[i:j]:
            t = line.strip()
            if re.match(r'^\.LBB\w+:', t):
                m = re.search(r'Header=(BB\w+) Depth=(\d+)', t)
                m2 = re.search(r'=>This (Inner )?Loop Header: Depth=(\d+)', t)
                if m2:
                    cur = (t.split(':')[0].lstrip('.L'), m2.group(2))
                elif m:
                    cur = (m.group(1), m.group(2))
                else:
                    cur = None
            k = classify(t)
            if k and cur:
                loops.setdefault(cur, collections.Counter())[k] += 1
            if t.startswith('s_endpgm'):
                break
        for (h, d), cc in loops.items():
            print('    loop %-10s depth %s  %s' % (h, d, dict(cc)))

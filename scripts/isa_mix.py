#!/usr/bin/env python3
"""Instruction mix of the kernels in a `hipcc -S --cuda-device-only` listing whose mangled name contains every word
given: floating-point VALU, other VALU, LDS, global memory, scalar, waits, barriers (static counts, straight-line
kernels: every instruction runs once per thread unless it sits in a loop).
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only pmx_colfft.hip -o /tmp/colfft.s
    python scripts/isa_mix.py /tmp/colfft.s colfft_round_kernelIfLi9ELb1"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
words = sys.argv[2:]
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

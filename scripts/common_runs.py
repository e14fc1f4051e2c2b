"""Development check (this container only: it reads /root/reference): runs of consecutive non-blank, non-comment lines
that a file of this repository has in common with a file of the reference, after whitespace normalisation.
usage: python scripts/common_runs.py pmesh_amd/pm.py /root/reference/pmesh/pm.py [minrun=5]"""
import re
import sys


def norm(path):
    out = []
    for no, line in enumerate(open(path), 1):
        t = re.sub(r'\s+', '', line.split('#')[0])
        if t and t not in ('"""', "'''"):
            out.append((no, t))
    return out


def runs(a, b, minrun):
    where = {}
    for j, (_, t) in enumerate(b):
        where.setdefault(t, []).append(j)
    found, i = [], 0
    while i < len(a):
        best = (0, -1)
        for j in where.get(a[i][1], ()):
            k = 0
            while i + k < len(a) and j + k < len(b) and a[i + k][1] == b[j + k][1]:
                k += 1
            if k > best[0]:
                best = (k, j)
        if best[0] >= minrun:
            found.append((best[0], a[i][0], a[i + best[0] - 1][0], b[best[1]][0], b[best[1] + best[0] - 1][0]))
            i += best[0]
        else:
            i += 1
    return found


if __name__ == '__main__':
    mine, ref = sys.argv[1], sys.argv[2]
    minrun = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    a, b = norm(mine), norm(ref)
    common = set(t for _, t in b)
    share = sum(1 for _, t in a if t in common) / max(1, len(a))
    print('%s vs %s: %.0f %% of %d lines occur in the reference' % (mine, ref, 100 * share, len(a)))
    for n, a0, a1, b0, b1 in runs(a, b, minrun):
        print('  run of %2d: %s:%d-%d  ==  ref:%d-%d' % (n, mine, a0, a1, b0, b1))

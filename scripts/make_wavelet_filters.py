#!/usr/bin/env python3
"""Scaling filters of the Daubechies (db6/12/20) and least-asymmetric (sym6/12/20) wavelets,
computed from their definition with mpmath (60 digits) and printed as the constants kept in
pmesh_amd/_tables.py.

    python scripts/make_wavelet_filters.py

Construction (I. Daubechies, Ten Lectures on Wavelets, ch. 6 and 8): the N-vanishing-moment
orthonormal scaling filter is h(z) ~ (1 + z)^N q(z) with |q|^2 fixed by
P(y) = sum_{k<N} C(N-1+k, k) y^k, y = (2 - z - 1/z) / 4; every root y of P gives a reciprocal
pair (z, 1/z) of which q takes one.  All roots inside the unit circle is the extremal-phase
"db" filter; the "sym" filters take the published least-asymmetric mix.  Which mix that is
(the 0/1 pattern per root group below, groups ordered by |z| and argument) was identified here
by regenerating the reference's lookup tables (tests/test_window.py::test_wavelet_tables_equal_reference).
"""
import sys
from math import comb

import mpmath

mpmath.mp.dps = 60

# one flag per root group, groups in canonical order: 0 = the root inside the unit circle
SELECTION = {}


def root_groups(N):
    coeffs = [mpmath.mpf(comb(N - 1 + k, k)) for k in range(N)]
    ys = mpmath.polyroots(coeffs[::-1], maxsteps=500, extraprec=200)
    pairs = []
    for y in ys:
        b = 2 - 4 * y
        d = mpmath.sqrt(b * b - 4)
        z1, z2 = (b + d) / 2, (b - d) / 2
        pairs.append((z1, z2) if abs(z1) < 1 else (z2, z1))
    # canonical order: by modulus of the inner root, then by |argument|, conjugates together
    pairs.sort(key=lambda p: (float(abs(p[0])), float(abs(mpmath.arg(p[0]))), float(mpmath.im(p[0]))))
    groups, used = [], [False] * len(pairs)
    for i, (zin, zout) in enumerate(pairs):
        if used[i]:
            continue
        used[i] = True
        g = [(zin, zout)]
        if abs(mpmath.im(zin)) > mpmath.mpf(10) ** -30:
            for j in range(i + 1, len(pairs)):
                if not used[j] and abs(pairs[j][0] - mpmath.conj(zin)) < mpmath.mpf(10) ** -20:
                    used[j] = True
                    g.append(pairs[j])
                    break
        groups.append(g)
    return groups


def scaling_filter(N, selection, reverse):
    poly = [mpmath.mpf(1)]
    for _ in range(N):                                   # (1 + z)^N, ascending powers
        poly = [a + b for a, b in zip(poly + [0], [0] + poly)]
    for g, s in zip(root_groups(N), selection):
        for zin, zout in g:
            z = zout if s else zin
            poly = [a - z * b for a, b in zip([0] + poly, poly + [0])]
    h = [mpmath.re(c) for c in poly]
    total = sum(h)
    h = [float(c * mpmath.sqrt(2) / total) for c in h]
    return h[::-1] if reverse else h


if __name__ == '__main__':
    import json
    spec = json.loads(sys.argv[1]) if len(sys.argv) > 1 else None
    if spec is None:
        print('usage: make_wavelet_filters.py \'{"db6": [6, [0,0,0], true], ...}\'')
        sys.exit(1)
    for name, (N, sel, rev) in spec.items():
        h = scaling_filter(N, sel, rev)
        print("    %r: %r," % (name, h))

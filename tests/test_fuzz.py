"""The random-configuration fuzzers of scripts/ as `-m gpu` tests: bounded subsets (about a minute altogether) of the
runs that found the real bugs of rounds 4 and 5 — every case compares the product's fast path with an independent
answer:

* scripts/paint_fuzz.py       tile-binned paint / readout kernels == the direct per-particle kernels (themselves pinned
                              to the oracle by tests/test_window.py), random geometries / windows / types / masses
* scripts/fft_fuzz.py         r2c / c2r of random mesh shapes == numpy.fft (the FFT oracle); deferred == eager
* scripts/halo_fuzz.py        halo merge left to r2c's row pass == the eagerly merged field, one rank
* scripts/halo_fuzz_slabs.py  the same on slab ranks (thread ranks of the one GPU)
* scripts/cycle_fuzz_ranks.py the whole PM cycle on P thread ranks (slabs, pencils, uneven blocks) == the one-rank cycle

Each runs in a process of its own (the fuzzers flip module-level switches of pmesh_amd.window / fft).  Case counts scale
with PMESH_AMD_FUZZ_SCALE (default 1; the long runs of the rounds are scale ~20); seeds differ from the scripts' defaults
so that the suite adds cases to what `scripts/*.py` alone covers.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE = float(os.environ.get('PMESH_AMD_FUZZ_SCALE', '1'))

# script, cases at scale 1, seed
FUZZERS = [
    ('paint_fuzz.py', 36, 606),
    ('fft_fuzz.py', 8, 66),
    ('halo_fuzz.py', 10, 6),
    ('halo_fuzz_slabs.py', 6, 16),
    ('cycle_fuzz_ranks.py', 6, 26),
]


@pytest.mark.gpu
@pytest.mark.parametrize('script,cases,seed', FUZZERS, ids=[f[0][:-3] for f in FUZZERS])
def test_fuzzer(script, cases, seed):
    n = max(1, int(round(cases * SCALE)))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', script), str(n), str(seed)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = '\n'.join((run.stdout + '\n' + run.stderr).splitlines()[-40:])
    assert run.returncode == 0, '%s %d %d failed:\n%s' % (script, n, seed, tail)

"""bench.py's output contract (the driver parses ONE JSON line from rank 0): every key it names, the metric string of
BASELINE.json verbatim, the roofline and cpu_baseline objects — on a small mesh so that the check takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_has_what_the_driver_reads():
    d = run_bench('--gpus', '1', '--steps', '3', '--warmup', '2', '--mesh', '128')
    metric = json.load(open(os.path.join(ROOT, 'BASELINE.json')))['metric']
    assert d['metric'] == metric
    assert d['unit'] == 'particles/s' and d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2
    assert d['higher_is_better'] is True and d['scaling'] in ('weak', 'strong') and d['vs_baseline'] is None
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] > 0 and abs(d['value'] - d['config']['particles'] / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] < 1 and 'traffic' in r
    # achieved = algorithmic bytes per launch over the kernel's measured time
    assert abs(r['achieved'] - r['algorithmic_bytes_per_particle'] * r['particles_per_launch'] / (r['ms_per_launch'] * 1e-3) / 1e9) \
        <= 1e-9 * r['achieved']
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['unit'] == 'particles/s' and c['sample']
    assert c['value'] is None or c['value'] > 0
    assert 'PMX_EXP' not in d['build_flags']
    # what the bin stage is, said in the line: a rebuild per step over positions moved by drift_cells
    assert d['bin_rebuilds_per_step'] == 1.0 and d['cold_plan'] is False and d['drift_cells'] > 0
    # the stages add up to the step (events on the stream)
    assert abs(sum(d['stages_ms'].values()) - d['ms_per_step']) <= 0.25 * d['ms_per_step']


def test_bench_other_forms_run():
    for args in (['--mesh', '128', '--window', 'tsc', '--dtype', 'f4', '--gradient', '0'],
                 ['--mesh', '128', '--out-field', '1'], ['--mesh', '128', '--data', 'clustered', '--window', 'pcs'],
                 ['--mesh', '128', '--cold-plan', '1'], ['--mesh', '128', '--drift', '2.0']):
        d = run_bench('--steps', '2', '--warmup', '1', '--no-cpu-baseline', *args)
        assert d['value'] > 0 and (d['bin_overflows'] == 0 or '--drift' in args)

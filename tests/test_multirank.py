"""The N > 1 path: one process per rank over torch.distributed.

CPU (`-m "not gpu"`): world sizes 2, 3 and 4 over gloo with the oracle double —
covers the layout exchange/gather, ghost handling, the slab FFT schedule
(pack -> all-to-all -> unpack, uneven blocks) and the full cycle.
GPU (`-m gpu`): the same cases on however many GPUs the box has (>= 2), HIP + RCCL.
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(nproc, backend, timeout=600):
    env = dict(os.environ)
    env['PMESH_MP_BACKEND'] = backend
    env['OMP_NUM_THREADS'] = '1'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % nproc,
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'mp_cases.py')]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-3000:] + '\n' + out.stderr[-6000:]
    return out.stdout


@pytest.mark.parametrize('nproc', [2, 3, 4])
def test_multirank_gloo(nproc):
    out = _launch(nproc, 'double')
    assert 'ok case_cycle on %d ranks' % nproc in out


@pytest.mark.gpu
def test_multirank_rccl():
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip('needs at least 2 GPUs (the 1-GPU box runs the single-rank suite)')
    out = _launch(min(n, 8), 'hip')
    assert 'ok case_cycle' in out


@pytest.mark.parametrize('nproc', [2, 3, 4, 8])
def test_multirank_threads(be, nproc):
    """The same cases with the ranks as threads of this process (tests/thread_comm.py).
    Under -m gpu this drives the real HIP kernels and the rocFFT stage plans of the slab
    FFT for P = 2, 4, 8 on a single device; under -m "not gpu" the oracle double."""
    from tests import mp_cases, thread_comm

    def body(comm):
        for case in mp_cases.CASES:
            case(be, comm)
            comm.Barrier()
    thread_comm.run_ranks(nproc, body)


@pytest.mark.gpu
@pytest.mark.parametrize('nproc,window,np_', [(8, 'cic', ''), (4, 'tsc', ''), (8, 'pcs', '2x4')])
def test_headline_cycle_distributed_equals_one_rank(nproc, window, np_):
    """The full-size cycle (512^3 mesh, 512^3 particles, BASELINE's multi-GPU metric) on P slab
    ranks — threads sharing the one GPU, real kernels, ghosts-only routing, pipelined transposes,
    fused transfer — or on a 2 x 4 pencil mesh with PCS (config 5's decomposition and window; the
    particles arrive in slabs of lattice ids, so most of them migrate) returns what the one-rank
    cycle returns, to 1e-11 of the result scale (measured 4e-15 / 1.4e-14).  The one-rank cycle is pinned to the oracle at this size by
    tests/test_binned.py::test_baseline_cycle_equals_oracle."""
    cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'mr_probe.py'), '--ranks', str(nproc), '--mesh', '512',
           '--steps', '1', '--warmup', '1', '--check', '1', '--window', window]
    if np_:
        cmd += ['--np', np_]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + '\n' + out.stderr[-4000:]
    assert 'vs one rank' in out.stdout

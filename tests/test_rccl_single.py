"""TorchComm over RCCL with a world of ONE rank (all a 1-GPU box can offer): every
collective the hot path issues is called with the argument types and device tensors the
multi-GPU run uses, so API misuse shows up here rather than on the 8-GPU node."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, %r)
import numpy, torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', world_size=1, rank=0)
from pmesh_amd.comm import TorchComm, default_comm
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
comm = default_comm()
assert isinstance(comm, TorchComm) and comm.size == 1
assert comm.allreduce(3.5) == 3.5
assert comm.allreduce(7) == 7
assert comm.allreduce(2.0, op='max') == 2.0
assert abs(comm.allreduce(1 + 2j) - (1 + 2j)) == 0
assert (comm.allreduce(numpy.arange(4.0)) == numpy.arange(4.0)).all()
assert comm.allgather({'a': 1}) == [{'a': 1}]
assert comm.bcast('x') == 'x'
assert (comm.alltoall_counts(numpy.array([5], dtype='int32')) == [5]).all()
dev = torch.device('cuda', 0)
a = torch.arange(12, dtype=torch.float64, device=dev).reshape(4, 3)
b = torch.empty_like(a)
comm.alltoallv(a, [4], b, [4]); assert torch.equal(a, b)
c = torch.empty(12, dtype=torch.float64, device=dev)
comm.alltoall(a.reshape(-1), c, [12], [12]); assert torch.equal(a.reshape(-1), c)
# the pipelined transposes: several asynchronous exchanges of slices of persistent buffers in
# flight while kernels run on the current stream; wait() orders the stream after each
W1 = torch.arange(4096, dtype=torch.float64, device=dev)
W2 = torch.zeros_like(W1)
works = [comm.alltoall(W1[o:o + 1024], W2[o:o + 1024], async_op=True) for o in (0, 1024, 2048)]
busy = (W1 * 2.0).sum()
for w in works:
    w.wait()
assert torch.equal(W2[:3072], W1[:3072]) and float(W2[3072:].abs().max()) == 0 and float(busy) == 4095 * 4096
comm.Barrier()
subs = comm.subgroups([[0], [0]])
assert subs[0].size == 1 and subs[1].size == 1
# the whole cycle through a TorchComm-backed ParticleMesh with a layout (the N > 1 code path
# of bench.py, degenerate to one rank)
N, L = 64, 1000.0
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', np=[1])
pos = torch.rand((N ** 3, 3), dtype=torch.float64, device=dev) * L
layout = pm.decompose(pos)
rho = pm.paint(pos, layout=layout)
assert abs(rho.csum() - N ** 3) < 1e-6
f = rho.r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=Transfer.dx1(0)).readout(pos, layout=layout)
assert bool(torch.isfinite(f).all())
dist.barrier(); dist.destroy_process_group()
print('rccl single ok')
'''


@pytest.mark.gpu
def test_torchcomm_on_rccl_world_of_one():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-c', SCRIPT % ROOT], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and 'rccl single ok' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]

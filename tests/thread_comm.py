"""P ranks as P threads of one process (TEST INFRASTRUCTURE).

Lets the multi-rank code paths — decompose/exchange/gather, the slab FFT schedule
with its rocFFT stage plans and pack kernels, the full cycle — run on a box with a
single GPU: every rank is a thread driving the same device, collectives are
implemented with barriers and device-to-device copies.  The product's communicator
for real runs is pmesh_amd.comm.TorchComm (RCCL).
"""
import threading

import numpy
import torch


class _Shared(object):
    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size
        self.lock = threading.Lock()


class ThreadComm(object):
    def __init__(self, shared, rank):
        self.shared = shared
        self.rank = rank
        self.size = shared.size

    def Barrier(self):
        self.shared.barrier.wait()

    def _exchange(self, obj):
        """all ranks deposit obj; returns the list of all deposits"""
        sh = self.shared
        sh.slots[self.rank] = obj
        sh.barrier.wait()
        allv = list(sh.slots)
        sh.barrier.wait()
        return allv

    def bcast(self, obj, root=0):
        return self._exchange(obj)[root]

    def allgather(self, obj):
        return self._exchange(obj)

    def allreduce(self, value, op='sum'):
        vals = self._exchange(value)
        f = {'sum': sum, 'max': max, 'min': min}[op]
        if isinstance(value, torch.Tensor):
            r = torch.stack(vals)
            return {'sum': r.sum(0), 'max': r.max(0)[0], 'min': r.min(0)[0]}[op]
        if isinstance(value, numpy.ndarray):
            a = numpy.stack(vals)
            return {'sum': a.sum(0), 'max': a.max(0), 'min': a.min(0)}[op]
        return f(vals)

    def alltoall_counts(self, sendcounts):
        allc = self._exchange(numpy.array(sendcounts))
        return numpy.array([allc[s][self.rank] for s in range(self.size)])

    def alltoallv(self, send, sendcounts, recv, recvcounts, async_op=False):
        self.alltoall(send, recv, [int(c) for c in sendcounts], [int(c) for c in recvcounts], rows=True)
        if async_op:
            from pmesh_amd.comm import _Done
            return _Done()

    def alltoall(self, send, recv, send_splits=None, recv_splits=None, rows=False, async_op=False):
        self._alltoall(send, recv, send_splits, recv_splits, rows)
        if async_op:
            from pmesh_amd.comm import _Done
            return _Done()

    def alltoall_views(self, send_views, recv_views, async_op=False):
        """one contiguous tensor per peer on either side (pmesh_amd.comm.TorchComm.alltoall_views)"""
        if send_views[0].is_cuda:
            torch.cuda.synchronize()
        allv = self._exchange(list(send_views))
        for s in range(self.size):
            recv_views[s].copy_(allv[s][self.rank].view(recv_views[s].shape))
        if send_views[0].is_cuda:
            torch.cuda.synchronize()
        self.shared.barrier.wait()
        if async_op:
            from pmesh_amd.comm import _Done
            return _Done()

    def _alltoall(self, send, recv, send_splits=None, recv_splits=None, rows=False):
        if send_splits is None:
            n = send.shape[0] // self.size
            send_splits = [n] * self.size
            recv_splits = [n] * self.size
        torch.cuda.synchronize() if send.is_cuda else None
        allv = self._exchange((send, list(send_splits)))
        off = 0
        for s in range(self.size):
            src, splits = allv[s]
            so = sum(splits[:self.rank])
            n = splits[self.rank]
            assert n == recv_splits[s], (n, recv_splits[s])
            recv[off:off + n].copy_(src[so:so + n])
            off += n
        torch.cuda.synchronize() if send.is_cuda else None
        self.shared.barrier.wait()


def _thread_subgroups(self, rank_lists):
    """sub-communicators over subsets of the thread ranks (shared state created by rank 0)"""
    key = tuple(tuple(r) for r in rank_lists)
    sh = self.shared
    if self.rank == 0:
        with sh.lock:
            if not hasattr(sh, 'subs'):
                sh.subs = {}
            sh.subs[key] = [_Shared(len(r)) for r in rank_lists]
    sh.barrier.wait()
    subs = sh.subs[key]
    sh.barrier.wait()
    return [ThreadComm(subs[i], r.index(self.rank)) if self.rank in r else None
            for i, r in enumerate(rank_lists)]


ThreadComm.subgroups = _thread_subgroups


def run_ranks(size, fn):
    """run fn(comm) on `size` threads; re-raises the first failure"""
    shared = _Shared(size)
    errors = []

    def work(rank):
        try:
            fn(ThreadComm(shared, rank))
        except BaseException as e:      # noqa
            errors.append((rank, e))
            shared.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)]
    if real:
        raise real[0][1]
    if errors:
        raise errors[0][1]

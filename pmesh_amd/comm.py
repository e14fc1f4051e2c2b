"""Communicators for the PM cycle.

The reference takes an mpi4py communicator everywhere (pmesh/pm.py:1311-1314,
pmesh/domain.py:92-123).  Here the unit of parallelism is one process per GPU and
the wire is RCCL over xGMI, reached through ``torch.distributed`` (backend
"nccl" *is* RCCL on ROCm; "gloo" on CPU tensors is used by the host-logic
tests).  Only the handful of operations the hot path needs are exposed, with
mpi4py-like names so the calling code reads like the reference's:

    rank, size, Barrier, bcast, allgather, allreduce           (python objects / scalars)
    alltoall_counts(sendcounts) -> recvcounts                  (domain.py:113)
    alltoallv(send, sendcounts, recv, recvcounts)              (domain.py:202, 278)
    alltoall(send, recv)                                       (PFFT's global transpose)

The reference brackets every exchange with two Barriers (domain.py:112-114, 199,
205, 274, 281); they are pure overhead on a stream-ordered device and are not
reproduced.
"""
import numpy
import torch


# ---- optional record of the collectives of the data path (bench.py --gpus N) ----------------
# trace(True) starts a list of [kind, bytes this rank sends to OTHER ranks, peers, start, end,
# overlapped]; start / end are CUDA events on the issuing stream (host clock readings for host
# tensors).  `overlapped` marks exchanges issued asynchronously: their window contains the kernels
# that ran underneath.
_trace = None


def trace(on=True):
    global _trace
    _trace = [] if on else None
    return _trace


def _stamp(t):
    if t.is_cuda:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(t.device))
        return e
    import time
    return time.perf_counter()


def trace_summary(records):
    """-> dict of totals over `records` (call after a device synchronisation)"""
    out = {'collectives': len(records), 'bytes_sent': 0, 'sync_ms': 0.0, 'overlapped_window_ms': 0.0,
           'max_link_GBs': 0.0, 'max_link_GBs_overlapped': 0.0}
    for kind, nbytes, peers, a, b, overlapped in records:
        ms = a.elapsed_time(b) if hasattr(a, 'elapsed_time') else 1e3 * (b - a)
        out['bytes_sent'] += nbytes
        out['overlapped_window_ms' if overlapped else 'sync_ms'] += ms
        if peers and ms > 0:
            # xGMI is point to point: one link per peer carries this rank's share for that peer.  (An overlapped
            # exchange's window also holds the kernels that ran underneath: its rate is a lower bound.)
            key = 'max_link_GBs_overlapped' if overlapped else 'max_link_GBs'
            out[key] = max(out[key], nbytes / peers / (ms * 1e-3) / 1e9)
    return out


class _Traced(object):
    """completes a trace record when the asynchronous exchange is waited for"""
    def __init__(self, work, rec, t):
        self.work, self.rec, self.t = work, rec, t

    def wait(self):
        r = self.work.wait()
        if self.rec[4] is None:          # (a handle may be waited for again by whoever reuses its buffers)
            self.rec[4] = _stamp(self.t)
        return r


class _Done(object):
    """handle of an exchange that has already happened"""
    def wait(self):
        return True


class SelfComm(object):
    """The single-process communicator (size 1): every collective is the identity."""
    rank = 0
    size = 1

    def Barrier(self):
        pass

    def bcast(self, obj, root=0):
        return obj

    def allgather(self, obj):
        return [obj]

    def allreduce(self, value, op='sum'):
        return value

    def alltoall_counts(self, sendcounts):
        return numpy.array(sendcounts, copy=True)

    def alltoallv(self, send, sendcounts, recv, recvcounts, async_op=False):
        recv.copy_(send)
        return _Done() if async_op else None

    def alltoall(self, send, recv, send_splits=None, recv_splits=None, async_op=False):
        recv.copy_(send)
        return _Done() if async_op else None

    def alltoall_views(self, send_views, recv_views, async_op=False):
        recv_views[0].copy_(send_views[0])
        return _Done() if async_op else None

    def subgroups(self, rank_lists):
        return [self for _ in rank_lists]


class TorchComm(object):
    """torch.distributed process group (RCCL for device tensors, gloo for host)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def _device_for_objects(self):
        backend = self._dist.get_backend(self.group)
        return torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')

    def Barrier(self):
        self._dist.barrier(group=self.group)

    def bcast(self, obj, root=0):
        box = [obj]
        src = root if self.group is None else self._dist.get_global_rank(self.group, root)
        self._dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def allgather(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj, group=self.group)
        return out

    def allreduce(self, value, op='sum'):
        """Sum (or max/min) of a python/numpy scalar, small numpy array, or a tensor."""
        ops = {'sum': self._dist.ReduceOp.SUM, 'max': self._dist.ReduceOp.MAX,
               'min': self._dist.ReduceOp.MIN}
        if isinstance(value, torch.Tensor):
            t = value.clone()
            self._dist.all_reduce(t, op=ops[op], group=self.group)
            return t
        a = numpy.asarray(value)
        dev = self._device_for_objects()
        if a.dtype.kind == 'c':
            t = torch.view_as_real(torch.from_numpy(numpy.atleast_1d(a).astype('c16'))).to(dev)
            self._dist.all_reduce(t, op=ops[op], group=self.group)
            r = torch.view_as_complex(t.cpu()).numpy()
        else:
            t = torch.from_numpy(numpy.atleast_1d(a).copy()).to(dev)
            self._dist.all_reduce(t, op=ops[op], group=self.group)
            r = t.cpu().numpy()
        if a.ndim == 0:
            return r.reshape(()).item() if not isinstance(value, numpy.generic) else r.reshape(())[()]
        return r

    def alltoall_counts(self, sendcounts):
        dev = self._device_for_objects()
        s = torch.as_tensor(numpy.asarray(sendcounts, dtype='i8')).to(dev)
        r = torch.empty_like(s)
        self._dist.all_to_all_single(r, s, group=self.group)
        return r.cpu().numpy()

    def alltoallv(self, send, sendcounts, recv, recvcounts, async_op=False):
        """rows of `send` (first axis) split by sendcounts -> rows of `recv`.  async_op: returns a handle whose
        wait() orders the current stream (RCCL) / the host (gloo) after the exchange — the particle exchanges of
        paint / readout run under the kernels of the caller's own particles"""
        rec = None
        if _trace is not None:
            row = send.element_size() * (send.numel() // max(1, send.shape[0])) if send.dim() else send.element_size()
            away = (int(numpy.sum(sendcounts)) - int(sendcounts[self.rank])) * row
            rec = ['alltoallv', away, self.size - 1, _stamp(send), None, bool(async_op)]
        w = self._dist.all_to_all_single(recv, send,
                                         output_split_sizes=[int(c) for c in recvcounts],
                                         input_split_sizes=[int(c) for c in sendcounts],
                                         group=self.group, async_op=async_op)
        if rec is not None:
            _trace.append(rec)
            if async_op:
                return _Traced(w, rec, send)
            rec[4] = _stamp(send)
        return w if async_op else None

    def alltoall(self, send, recv, send_splits=None, recv_splits=None, async_op=False):
        """async_op: returns a handle whose wait() orders the *current stream* (RCCL) / the host
        (gloo) after the exchange; the buffers must stay untouched until then"""
        rec = None
        if _trace is not None:
            total = send.numel() * send.element_size()
            if send_splits is None:
                away = total - total // self.size
            else:
                row = total // max(1, send.shape[0])
                away = (int(sum(send_splits)) - int(send_splits[self.rank])) * row
            rec = ['alltoall', away, self.size - 1, _stamp(send), None, bool(async_op)]
        if send_splits is None:
            w = self._dist.all_to_all_single(recv, send, group=self.group, async_op=async_op)
        else:
            w = self._dist.all_to_all_single(recv, send, output_split_sizes=list(recv_splits),
                                             input_split_sizes=list(send_splits), group=self.group,
                                             async_op=async_op)
        if rec is not None:
            _trace.append(rec)
            if async_op:
                return _Traced(w, rec, send)
            rec[4] = _stamp(send)
        return w if async_op else None


class _Staged(object):
    """an exchange through one staging buffer (backends without the list form of all-to-all): the
    pieces are copied to their places when the exchange is waited for"""
    def __init__(self, work, staging, recv_views):
        self.work, self.staging, self.recv_views = work, staging, recv_views

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if self.staging is not None:
            o = 0
            for v in self.recv_views:
                n = v.numel()
                v.copy_(self.staging[o:o + n].view(v.shape))
                o += n
            self.staging = None          # (waiting again is a no-op)
        return True


def _torch_alltoall_views(self, send_views, recv_views, async_op=False):
    """all-to-all of one contiguous tensor per peer on either side (views into larger arrays: row
    ranges that are contiguous by themselves but not next to each other — the pieces of a chunked
    pencil transpose).  RCCL takes the lists as they are (grouped send / recv); gloo has no list
    form: the pieces travel through one staging buffer on either side."""
    rec = None
    if _trace is not None:
        away = sum(v.numel() * v.element_size() for i, v in enumerate(send_views) if i != self.rank)
        rec = ['alltoall', away, self.size - 1, _stamp(send_views[0]), None, bool(async_op)]
        _trace.append(rec)
    if self._dist.get_backend(self.group) == 'nccl':
        w = self._dist.all_to_all(list(recv_views), list(send_views), group=self.group, async_op=async_op)
    else:
        send = torch.cat([v.reshape(-1) for v in send_views])
        staging = torch.empty(sum(v.numel() for v in recv_views), dtype=send.dtype, device=send.device)
        work = self._dist.all_to_all_single(staging, send, output_split_sizes=[v.numel() for v in recv_views],
                                            input_split_sizes=[v.numel() for v in send_views], group=self.group,
                                            async_op=async_op)
        w = _Staged(work if async_op else None, staging, recv_views)
        if not async_op:
            w.wait()
    if rec is not None:
        if async_op:
            return _Traced(w, rec, send_views[0])
        rec[4] = _stamp(send_views[0])
    return w if async_op else None


TorchComm.alltoall_views = _torch_alltoall_views


def _torch_subgroups(self, rank_lists):
    """one communicator per rank list (every rank creates every group, in the same order, as
    torch.distributed requires); entries this rank is not a member of are None"""
    out = []
    for ranks in rank_lists:
        granks = ranks if self.group is None else [self._dist.get_global_rank(self.group, r) for r in ranks]
        g = self._dist.new_group(ranks=granks)
        out.append(TorchComm(group=g) if self.rank in ranks else None)
    return out


TorchComm.subgroups = _torch_subgroups


_self_comm = SelfComm()
_world_comm = {}


def default_comm():
    """WORLD if torch.distributed is initialised, else the single-process comm.  The same
    object every time (ParticleMesh recognises a communicator it has plans for by identity)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            key = id(dist.group.WORLD)
            if key not in _world_comm:
                _world_comm.clear()
                _world_comm[key] = TorchComm()
            return _world_comm[key]
    except Exception:
        pass
    return _self_comm

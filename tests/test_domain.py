"""GridND.decompose / Layout.exchange / Layout.gather of pmesh_amd.domain.

Golden: tests/golden/decompose.npz holds counts and indices produced by the
reference's own domain.py + compiled _domain.gridnd_fill for 1/2/3/4/8 ranks
(slabs, pencils, a 2x2x2 grid, uneven and degenerate edges, a custom
DomainAssign), periodic and not, several smoothings.  Index work is compared
bit-exact in both modes (`-m gpu`: HIP kernels; otherwise: host logic + oracle).
Known answers restate pmesh/tests/test_domain.py.
"""
import numpy
import pytest
import torch
from numpy.testing import assert_array_equal, assert_allclose

from pmesh_amd import domain


class FakeComm(object):
    """size-P communicator with no peers: enough for decompose, which only
    needs comm.size / comm.rank and the count exchange of Layout.__init__."""
    def __init__(self, size, rank=0):
        self.size, self.rank = size, rank

    def alltoall_counts(self, sendcounts):
        return numpy.array(sendcounts)

    def allgather(self, x):
        return [x] * self.size

    def allreduce(self, x, op='sum'):
        return x


def _cases(g):
    return sorted(k[:-len('/counts')] for k in g.files if k.endswith('/counts'))


def test_golden_decompose(be, golden):
    g = golden['decompose']
    n = 0
    for tag in _cases(g):
        cname, per, sm, sc, ptag = tag.split('/')
        edges = [g['%s/edges%d' % (cname, d)] for d in range(3)]
        P = int(g['%s/nranks' % cname][0])
        grid = domain.GridND(edges, comm=FakeComm(P), periodic=(per == 'per'),
                             DomainAssign=g['%s/assign' % cname])
        assert_array_equal(grid.DomainDegenerate, g['%s/degenerate' % cname])
        pos = g['pos'] if ptag == 'f8' else g['pos_f4']
        layout = grid.decompose(pos, smoothing=eval(sm[2:]), _scale=float(sc[2:]))
        assert_array_equal(layout.sendcounts, g[tag + '/counts'], err_msg=tag)
        assert layout.sendcounts.dtype == numpy.dtype('int32')
        assert layout.indices.dtype == torch.int32
        assert_array_equal(layout.indices.cpu().numpy(), g[tag + '/indices'], err_msg=tag)
        n += 1
    assert n > 100


def test_decompose_transform_callable(be, golden):
    """the public signature takes a callable transform (domain.py:561-585)"""
    g = golden['decompose']
    edges = [g['slab4/edges%d' % d] for d in range(3)]
    grid = domain.GridND(edges, comm=FakeComm(4), periodic=True)
    a = grid.decompose(g['pos'], smoothing=1.0, transform=lambda x: 0.5 * x)
    assert_array_equal(a.sendcounts, g['slab4/per/sm1.0/sc0.5/f8/counts'])
    assert_array_equal(a.indices.cpu().numpy(), g['slab4/per/sm1.0/sc0.5/f8/indices'])


def test_empty_and_single(be):
    grid = domain.GridND([[0, 1, 2], [0, 2]], comm=FakeComm(2))
    layout = grid.decompose(numpy.empty((0, 2)), smoothing=0)
    assert layout.sendlength == 0 and len(layout.indices) == 0
    assert_array_equal(layout.sendcounts, [0, 0])


def test_exchange_gather_single_rank(be):
    """On one rank the layout still creates ghosts for a periodic self-overlap and
    gather reduces them (test_domain.py:243-266 pattern)."""
    from pmesh_amd.comm import SelfComm
    grid = domain.GridND([[0, 2], [0, 2]], comm=SelfComm(), periodic=True)
    pos = numpy.array(list(numpy.ndindex((2, 2))), dtype='f8')
    layout = grid.decompose(pos, smoothing=1)
    npos = layout.exchange(pos)
    assert_array_equal(npos, pos)
    mass = numpy.array([0., 1., 2., 3.])
    nmass = layout.exchange(mass)
    assert_array_equal(layout.gather(nmass, mode='sum'), mass)
    assert_array_equal(layout.gather(nmass, mode='any'), mass)
    assert_array_equal(layout.gather(nmass, mode='local'), mass)
    assert_array_equal(layout.gather(nmass, mode='mean'), mass)
    assert_array_equal(layout.gather(nmass, mode=numpy.fmax), mass)
    # integer payloads and 2-d payloads are exchanged too (test_domain.py:78-90)
    assert_array_equal(layout.exchange([0, 1, 2, 3]), [0, 1, 2, 3])
    a, b = layout.exchange(pos, mass)
    assert_array_equal(a, pos) and assert_array_equal(b, mass)
    with pytest.raises(ValueError):
        layout.exchange(numpy.zeros(3))
    with pytest.raises(ValueError):
        layout.gather(numpy.zeros(3))


def test_uniform_and_loadbalance(be):        # test_domain.py:9-27, 309-336
    for P, shape in ((4, (1, 2, 2)), (3, (1, 3, 1)), (2, (1, 2, 1)), (1, (1, 1, 1))):
        dcop = domain.GridND.uniform(BoxSize=[1, 2, 2], comm=FakeComm(P), periodic=True)
        assert_array_equal(dcop.shape, shape)
    dcop = domain.GridND([[0, 1, 2, 3, 4], [0, 2, 4]], comm=FakeComm(4), periodic=True)
    dcop.loadbalance([5, 4, 9, 3, 15, 6, 8, 1])
    assert not any(dcop.DomainAssign - [3, 2, 1, 1, 0, 3, 2, 3])
    dcop = domain.GridND([[0, 1, 2, 3], [0, 3]], comm=FakeComm(4), periodic=True)
    dcop.loadbalance([10, 6, 12])
    assert not any(dcop.DomainAssign - [0, 1, 2])


def test_load_and_isprimary(be):             # test_domain.py:270-305
    from pmesh_amd.comm import SelfComm
    dcop = domain.GridND([[0, 1, 2], [0, 2]], comm=SelfComm(), periodic=True)
    pos = numpy.array(list(numpy.ndindex((3, 6, 1))), dtype='f8')
    assert sum(dcop.load(pos, gamma=1)) == len(pos)
    pos = numpy.array(list(numpy.ndindex((6, 6, 1))), dtype='f8') - 2
    layout = dcop.decompose(pos, smoothing=1.5)
    npos = layout.exchange(pos)
    assert dcop.isprimary(npos).sum() >= len(pos)


def test_bincountv(be):
    rs = numpy.random.RandomState(2)
    idx = rs.randint(0, 20, size=100)
    w = rs.normal(size=(100, 3))
    got = domain.bincountv(idx, w, minlength=20)
    want = numpy.stack([numpy.bincount(idx, w[:, c], minlength=20) for c in range(3)], axis=-1)
    assert_allclose(got, want, rtol=0, atol=1e-13)

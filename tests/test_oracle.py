"""Pin the CPU oracle (oracle/pmesh_oracle.c) to the reference.

Three anchors:
  * the golden vectors generated from the compiled reference extensions and the
    reference's window.py / domain.py (tests/golden/*.npz, make_golden.py);
  * the inline known answers of the reference's pmesh/tests/test_window.py;
  * where oracle/_ref exists (the reference's _window_imp.c compiled from where
    it lies), a direct bit-for-bit comparison on fresh random inputs.
All comparisons here are exact (array_equal): the oracle restates the reference's
operation order and is built without FP contraction.
"""
import numpy
import pytest
from numpy.testing import assert_array_equal, assert_allclose, assert_almost_equal

TUNED = {'nnb': 'tunednnb', 'cic': 'tunedcic', 'tsc': 'tunedtsc', 'pcs': 'tunedpcs'}
GENERIC = {'nearest': 'nearest', 'linear': 'linear', 'quadratic': 'quadratic', 'cubic': 'cubic'}
ALL = dict(TUNED, **GENERIC)


def _dd(tag):
    return None if tag == 'n' else int(tag)


@pytest.mark.parametrize('name', sorted(ALL))
def test_golden_case_A(oracle, golden, name):
    g = golden['window']
    W = oracle.Window(ALL[name])
    aff = oracle.Affine(3, scale=g['A/scale'], translate=g['A/translate'], period=g['A/period'])
    for dt in ('f8', 'f4'):
        for tag in ('n', '0', '1', '2'):
            key = 'A/%s/%s/%s' % (name, dt, tag)
            real = numpy.zeros(g[key + '/paint'].shape, dtype=dt)
            W.paint(real, g['A/pos'], mass=g['A/mass'], diffdir=_dd(tag), transform=aff)
            assert_array_equal(real, g[key + '/paint'], err_msg=key)
            v = W.readout(g['A/field'].astype(dt), g['A/pos'], diffdir=_dd(tag), transform=aff)
            assert_array_equal(v, g[key + '/readout'], err_msg=key)


@pytest.mark.parametrize('name', sorted(ALL))
def test_golden_case_B_f4(oracle, golden, name):
    g = golden['window']
    W = oracle.Window(ALL[name])
    real = numpy.zeros(g['B/%s/paint' % name].shape, dtype='f4')
    W.paint(real, g['B/pos'], mass=2.5)
    assert_array_equal(real, g['B/%s/paint' % name])
    o = numpy.zeros(len(g['B/pos']), dtype='f4')
    W.readout(g['B/field'], g['B/pos'], out=o)
    assert_array_equal(o, g['B/%s/readout' % name])


@pytest.mark.parametrize('name', sorted(ALL))
def test_golden_case_C_lowdim(oracle, golden, name):
    g = golden['window']
    W = oracle.Window(ALL[name])
    for tag in ('n', '0', '1'):
        aff = oracle.Affine(2, scale=[1.0, 0.5], translate=[0.5, -1], period=[9, 7])
        key = 'C/%s/2/%s' % (name, tag)
        real = numpy.zeros((9, 7))
        W.paint(real, g['C/pos2'], mass=g['C/mass'], diffdir=_dd(tag), transform=aff)
        assert_array_equal(real, g[key + '/paint'])
        assert_array_equal(W.readout(g['C/field2'], g['C/pos2'], diffdir=_dd(tag), transform=aff),
                           g[key + '/readout'])
    for tag in ('n', '0'):
        aff = oracle.Affine(1, scale=[0.9], translate=[0.1], period=[11])
        key = 'C/%s/1/%s' % (name, tag)
        real = numpy.zeros((11,))
        W.paint(real, g['C/pos1'], mass=g['C/mass'][:100], diffdir=_dd(tag), transform=aff)
        assert_array_equal(real, g[key + '/paint'])
        assert_array_equal(W.readout(g['C/field1'], g['C/pos1'], diffdir=_dd(tag), transform=aff),
                           g[key + '/readout'])


@pytest.mark.parametrize('name', sorted(ALL))
def test_golden_case_D_hsml_resize(oracle, golden, name):
    g = golden['window']
    W = oracle.Window(ALL[name])
    aff = oracle.Affine(3, period=12)
    real = numpy.zeros((12, 12, 12))
    W.paint(real, g['D/pos'], hsml=g['D/hsml'], transform=aff)
    assert_array_equal(real, g['D/%s/paint' % name])
    assert_array_equal(W.readout(g['D/field'], g['D/pos'], hsml=g['D/hsml'], transform=aff),
                       g['D/%s/readout' % name])
    assert_array_equal(W.readout(g['D/field'], g['D/pos'], hsml=g['D/hsml'], transform=aff, diffdir=1),
                       g['D/%s/readout_g1' % name])
    W6 = W.resize(6)
    assert_array_equal([W6.support, W6.nativesupport], g['D/%s/resize6/support' % name])
    real = numpy.zeros((12, 12, 12))
    W6.paint(real, g['D/pos'], transform=aff)
    assert_array_equal(real, g['D/%s/resize6/paint' % name])
    assert_array_equal(W6.readout(g['D/field'], g['D/pos'], transform=aff),
                       g['D/%s/resize6/readout' % name])


@pytest.mark.parametrize('name', sorted(TUNED))
def test_golden_case_E_F_G(oracle, golden, name):
    g = golden['window']
    W = oracle.Window(TUNED[name])
    for dt in ('f8', 'f4'):
        aff = oracle.Affine(3, period=8)
        real = numpy.zeros((8, 8, 8), dtype=dt)
        W.paint(real, g['E/pos'], mass=g['E/mass'], transform=aff)
        assert_array_equal(real, g['E/%s/%s/paint' % (name, dt)])
        assert_array_equal(W.readout(g['E/field'].astype(dt), g['E/pos'], transform=aff),
                           g['E/%s/%s/readout' % (name, dt)])
    N, L = int(g['F/N'][0]), float(g['F/L'][0])
    aff = oracle.Affine(3, scale=1.0 * N / L, period=N)
    real = numpy.zeros((N, N, N))
    W.paint(real, g['F/pos'], transform=aff)
    assert_array_equal(real, g['F/%s/paint' % name])
    assert_array_equal(W.readout(g['F/field'], g['F/pos'], transform=aff), g['F/%s/readout' % name])
    big = numpy.zeros((18, 12))
    W.paint(big[::3, ::2], g['G/pos'], transform=oracle.Affine(2, period=[6, 6]))
    assert_array_equal(big, g['G/%s/big' % name])
    cplx = numpy.zeros((6, 6), dtype='c16')
    W.paint(cplx, g['G/pos'], transform=oracle.Affine(2, period=[6, 6]))
    assert_array_equal(cplx, g['G/%s/complex' % name])


@pytest.mark.parametrize('name', sorted(ALL))
def test_golden_fwindow(oracle, golden, name):
    g = golden['window']
    W = oracle.Window(ALL[name])
    assert_array_equal([W.support, W.nativesupport], g['W/%s/support' % name])
    assert_array_equal(W.get_fwindow(g['W/w']), g['W/%s/fwindow' % name])
    assert_array_equal(W.resize(6).get_fwindow(g['W/w']), g['W/%s/resize6/fwindow' % name])


# ---- the reference's inline known answers (pmesh/tests/test_window.py) ------

def test_ref_known_answers(oracle):
    CIC = oracle.Window('tunedcic')
    TSC = oracle.Window('tunedtsc')
    Affine = oracle.Affine
    pos4 = [[0., 0.], [1., 1.], [2., 2.], [3., 3.]]
    real = numpy.zeros((4, 4))
    CIC.paint(real, pos4)                                   # test_unweighted :11
    assert_array_equal(real, numpy.eye(4))
    real = numpy.zeros((4, 4))
    CIC.paint(real, pos4, mass=numpy.array([0., 1., 2., 3.]))   # test_weighted :27
    assert_array_equal(real, numpy.diag([0., 1, 2, 3]))
    wcic = oracle.Window('linear', 4)                       # test_wide :43
    real = numpy.zeros(4)
    wcic.paint(real, [[1.5]])
    assert_almost_equal(real, [0.125, 0.375, 0.375, 0.125])
    real = numpy.zeros(4)
    wcic.paint(real, [[1.51]])
    assert_almost_equal(real, [0.1225, 0.3725, 0.3775, 0.1275])
    real = numpy.zeros(4)
    wcic.paint(real, [[1.5]], diffdir=0)
    assert_almost_equal(real, [-0.25, -0.25, 0.25, 0.25])
    for p in ([[-.5, -.5]], [[-.5, .5]], [[-.5, 1.5]]):     # test_wrap :60
        real = numpy.zeros((2, 2))
        CIC.paint(real, p, transform=Affine(2, period=2))
        assert_array_equal(real, numpy.full((2, 2), 0.25))
    real = numpy.zeros((2, 2))                              # test_translate :89
    CIC.paint(real, [[1., 0]], transform=Affine(2, translate=[-1, 0]))
    assert_array_equal(real, [[1., 0.], [0., 0.]])
    real = numpy.zeros((2, 2))                              # test_scale :118
    CIC.paint(real, [[10., 0]], transform=Affine(2, translate=[-1, 0], scale=0.1))
    assert_almost_equal(real, [[1., 0.], [0, 0.]])
    real = numpy.zeros(10)                                  # test_scale_hsml :127
    CIC.paint(real, [[50., 0]], hsml=1., transform=Affine(1, translate=[0], scale=0.1))
    assert_array_equal(real, numpy.eye(10)[5])
    real = numpy.zeros((20, 20))[::10, ::10]                # test_strides :145
    CIC.paint(real, [[1., 0]])
    assert_array_equal(real, [[0, 0], [1, 0]])
    real = numpy.zeros((2, 4))                              # test_anisotropic :155
    CIC.paint(real, [[0., 0], [1., 0], [0., 1], [0., 2], [0., 3]])
    assert_array_equal(real, [[1, 1, 1, 1], [1, 0, 0, 0]])
    real = numpy.zeros((2, 2))                              # test_diff :169
    CIC.paint(real, [[0.5, 0]], diffdir=0)
    assert_array_equal(real, [[-1, 0], [1, 0]])
    real = numpy.zeros((2, 2))
    CIC.paint(real, [[0, 0.5]], diffdir=1)
    assert_array_equal(real, [[-1, 1], [0, 0]])
    real = numpy.zeros((4, 4))                              # test_nearest :188
    oracle.Window('nearest').paint(real, [[1.2, 1.2]])
    e = numpy.zeros((4, 4)); e[1, 1] = 1
    assert_allclose(real, e, atol=1e-5)
    real = numpy.zeros(4)                                   # test_tsc :222
    TSC.paint(real, [[1.5]])
    assert_array_equal(real, [0, 0.5, 0.5, 0])
    real = numpy.zeros(4)
    TSC.paint(real, [[1.8]])
    assert_almost_equal(real, [0., 0.245, 0.71, 0.045])
    real = numpy.zeros(5)
    TSC.paint(real, [[2.]])
    assert_array_equal(real, [0, 0.125, 0.75, 0.125, 0])
    real = numpy.zeros(5)
    TSC.paint(real, [[0.]], transform=Affine(1, period=5))
    assert_array_equal(real, [0.75, 0.125, 0, 0, 0.125])
    real = numpy.zeros(6)                                   # test_cubic :253
    oracle.Window('cubic').paint(real, [[2.5]])
    assert_allclose(real, [0., 0.02083333, 0.47916667, 0.47916667, 0.02083333, 0.], rtol=1e-6)
    r1 = numpy.zeros(10); r2 = numpy.zeros(10)              # test_cubic_hsml :264
    oracle.Window('cubic').paint(r1, [[4.5]], hsml=2.0)
    oracle.Window('cubic').resize(8).paint(r2, [[4.5]], hsml=1.0)
    assert_array_equal(r1, r2)
    assert_allclose(CIC.get_fwindow([0, 2 * numpy.pi]), [1, 0.0], atol=1e-9)  # test_compensation :362


def test_ref_tuned_equals_generic(oracle):
    """test_cic_tuned :311 / test_tsc_tuned :332 restated on the oracle."""
    Affine = oracle.Affine
    pos = [[1.1, 1.3, 2.5]]
    for d in (None, 0, 1, 2):
        a = numpy.zeros((4, 4, 4)); b = numpy.zeros((4, 4, 4))
        oracle.Window('tunedcic').paint(a, pos, diffdir=d)
        oracle.Window('linear').paint(b, pos, diffdir=d)
        assert_array_equal(a, b)
    aff = Affine(3, translate=[2, 1, 2], scale=[0.5, 2.0, 1.1], period=[8, 8, 8])
    field = numpy.random.RandomState(1234).uniform(size=(8, 8, 8))
    pos = [[1.1, 1.3, 2.9]]
    for d in (None, 0, 1, 2):
        a = numpy.zeros((8, 8, 8)); b = numpy.zeros((8, 8, 8))
        oracle.Window('tunedtsc').paint(a, pos, diffdir=d, transform=aff)
        oracle.Window('quadratic').paint(b, pos, diffdir=d, transform=aff)
        assert_array_equal(a, b)
        assert_array_equal(oracle.Window('tunedtsc').readout(field, pos, diffdir=d, transform=aff),
                           oracle.Window('quadratic').readout(field, pos, diffdir=d, transform=aff))


def test_seeded_spot_checks(oracle):
    """values captured from the compiled reference."""
    N = 16
    pos = numpy.random.RandomState(42).uniform(-4, N + 4, (1000, 3))
    mass = numpy.random.RandomState(43).uniform(.5, 1.5, 1000)
    field = numpy.random.RandomState(1).normal(size=(N, N, N))
    aff = oracle.Affine(3, period=N)
    table = {
        'tunednnb': (1012.1109673078859, 1450.653214584491, 0.0, -28.198197859843507,
                     -0.32674455138175723, 0.0, 0.0),
        'tunedcic': (1012.1109673078857, 659.9045204032627, 0.09231457746691686, -9.248091071042731,
                     -0.17959665375440456, 12.143797481925793, -0.01911816172483799),
        'tunedtsc': (1012.1109673078859, 512.0239547225945, 0.19164168176472451, -3.337725027268794,
                     -0.19651225335181405, -1.6368611525947463, 0.18479123839054595),
        'tunedpcs': (1012.1109673078856, 449.01015174824806, 0.21962417717466742, 1.0607633154994094,
                     -0.1745896629115248, -1.0130773906648676, 0.2247239749681089),
    }
    for kind, exp in table.items():
        W = oracle.Window(kind)
        r = numpy.zeros((N, N, N))
        W.paint(r, pos, mass=mass, transform=aff)
        v = W.readout(field, pos, transform=aff)
        g = W.readout(field, pos, transform=aff, diffdir=1)
        got = (r.sum(), (r ** 2).sum(), r[3, 5, 7], v.sum(), v[0], g.sum(), g[0])
        assert got == exp, kind


def test_against_compiled_reference(oracle):
    """oracle == oracle/_ref (the reference's _window_imp.c) on fresh inputs."""
    if not oracle.have_ref():
        pytest.skip('oracle/_ref not built (reference sources absent)')
    rs = numpy.random.RandomState(5)
    for kind in sorted(set(ALL.values())):
        for dt in ('f8', 'f4'):
            for nd, shape in ((3, (7, 5, 9)), (2, (6, 11)), (1, (13,))):
                pos = rs.uniform(-10, 25, size=(500, nd)).astype(rs.choice(['f4', 'f8']))
                mass = rs.uniform(0, 2, size=500)
                hsml = rs.uniform(0.3, 2.5, size=500) if rs.rand() < 0.5 else None
                field = rs.normal(size=shape).astype(dt)
                period = [s + int(rs.randint(0, 3)) for s in shape] if rs.rand() < 0.7 else [0] * nd
                scale = rs.uniform(0.3, 1.5, size=nd)
                transl = rs.uniform(-3, 3, size=nd)
                for d in [None] + list(range(nd)):
                    outs = []
                    for which in ('oracle', 'ref'):
                        W = oracle.Window(kind, which=which)
                        aff = oracle.Affine(nd, scale=scale, translate=transl, period=period)
                        real = numpy.zeros(shape, dtype=dt)
                        W.paint(real, pos, mass=mass, hsml=hsml, diffdir=d, transform=aff)
                        v = W.readout(field, pos, hsml=hsml, diffdir=d, transform=aff)
                        outs.append((real, v))
                    assert_array_equal(outs[0][0], outs[1][0], err_msg='%s %s %d' % (kind, dt, nd))
                    assert_array_equal(outs[0][1], outs[1][1], err_msg='%s %s %d' % (kind, dt, nd))


# ---- decomposition ----------------------------------------------------------

def _decompose_cases(g):
    tags = set()
    for k in g.files:
        if k.endswith('/counts'):
            tags.add(k[:-len('/counts')])
    return sorted(tags)


def test_golden_decompose(oracle, golden):
    g = golden['decompose']
    ncases = 0
    for tag in _decompose_cases(g):
        cname, per, sm, sc, ptag = tag.split('/')
        edges = [g['%s/edges%d' % (cname, d)] for d in range(3)]
        P = int(g['%s/nranks' % cname][0])
        grid = oracle.GridSpec(edges, P, periodic=(per == 'per'), DomainAssign=g['%s/assign' % cname])
        assert_array_equal(grid.DomainDegenerate, g['%s/degenerate' % cname])
        smoothing = eval(sm[2:])
        pos = g['pos'] if ptag == 'f8' else g['pos_f4']
        counts, indices = oracle.decompose(grid, pos, smoothing, scale=float(sc[2:]))
        assert_array_equal(counts, g[tag + '/counts'], err_msg=tag)
        assert_array_equal(indices, g[tag + '/indices'], err_msg=tag)
        ncases += 1
    assert ncases > 100


def test_take_scatter(oracle):
    rs = numpy.random.RandomState(3)
    data = rs.normal(size=(50, 3))
    idx = rs.randint(0, 50, size=200).astype('int32')
    assert_array_equal(oracle.take_rows(data, idx), data.take(idx, axis=0))
    vals = rs.normal(size=200)
    out = oracle.scatter_add(vals, idx, 50)
    assert_array_equal(out, numpy.bincount(idx, vals, minlength=50))
    vals4 = vals.astype('f4')
    out4 = oracle.scatter_add(vals4, idx, 50)
    assert_array_equal(out4, numpy.bincount(idx, vals4, minlength=50).astype('f4'))


# ---- the PM cycle -----------------------------------------------------------

def test_golden_cycle(oracle, golden):
    g = golden['cycle16']
    N, L = int(g['N'][0]), float(g['L'][0])
    pos = g['pos']
    transfers = {
        'dx1_0': oracle.make_transfer(laplace_pow=-1, grad_dir=0, grad_kind=0),
        'force_2': oracle.make_transfer(laplace_pow=-1, grad_dir=2, grad_kind=1),
        'pot': oracle.make_transfer(amplitude=-1.0, laplace_pow=-1),
    }
    for name, kind in TUNED.items():
        for tname, t in transfers.items():
            real, ck, back, out = oracle.pm_cycle(N, L, pos, kind=kind, transfer=t)
            assert_array_equal(real, g['%s/paint' % name])
            assert_allclose(back, g['%s/%s/c2r' % (name, tname)], rtol=0,
                            atol=1e-12 * abs(g['%s/%s/c2r' % (name, tname)]).max())
            assert_allclose(out, g['%s/%s/readout' % (name, tname)], rtol=0,
                            atol=1e-12 * abs(g['%s/%s/readout' % (name, tname)]).max())


def test_synthetic_inputs(oracle):
    pos = oracle.synth_uniform(8, 1000.0)
    assert pos.shape == (512, 3)
    q = (numpy.indices((8, 8, 8)).reshape(3, -1).T + 0.5) * 125.0
    assert abs(pos - q).max() <= 0.4 * 125.0
    assert abs(pos - q).std() > 0.1 * 125.0
    # chunks are consistent with the whole
    part = oracle.synth_uniform(8, 1000.0, g0=100, npart=50)
    assert_array_equal(part, pos[100:150])
    modes = oracle.zeldovich_modes(8, 1000.0)
    c = oracle.synth_clustered(8, 1000.0, modes)
    assert (c >= 0).all() and (c < 1000.0).all()
    d = (c - q + 500.0) % 1000.0 - 500.0
    rms = numpy.sqrt((d ** 2).sum(axis=1).mean())
    assert 1.5 * 125 < rms < 4.5 * 125


@pytest.mark.parametrize('kind', ['nearest', 'linear', 'quadratic', 'cubic', 'tunednnb', 'tunedcic', 'tunedtsc', 'tunedpcs'])
@pytest.mark.parametrize('shape,period', [((6, 5, 7, 4), (6, 5, 7, 0)), ((4, 3, 5, 4, 3), (4, 3, 5, 4, 3))])
def test_more_than_three_dimensions_equal_compiled_reference(oracle, kind, shape, period):
    """pmo_paint_nd / pmo_readout_nd (meshes of 4 .. 8 dimensions: the reference's generic product, its tuned kernels
    stop at three, _window_imp.c:486-520) == the reference's own _window_imp.c compiled where it lies (oracle/_ref),
    bit for bit: anisotropic affine, a non-periodic axis, gradients, per-particle hsml, f4 / f8 canvases."""
    if not oracle.have_ref():
        pytest.skip('oracle/_ref (compiled reference) is not available')
    nd = len(shape)
    rs = numpy.random.RandomState(31 + nd)
    pos = rs.uniform(-3, 9, size=(300, nd))
    mass = rs.uniform(0.5, 1.5, size=300)
    hs = rs.uniform(0.8, 1.6, size=300)
    aff = oracle.Affine(nd, scale=[1.0, 0.5, 1.5, 1.0, 0.7][:nd], translate=[0.2, 0, -0.3, 0.1, 0][:nd], period=period)
    for dt in ('f8', 'f4'):
        for diffdir in (None, 2):
            a, b = numpy.zeros(shape, dtype=dt), numpy.zeros(shape, dtype=dt)
            oracle.Window(kind).paint(a, pos, mass=mass, transform=aff, diffdir=diffdir)
            oracle.Window(kind, which='ref').paint(b, pos, mass=mass, transform=aff, diffdir=diffdir)
            assert_array_equal(a, b)
            assert_array_equal(oracle.Window(kind).readout(a, pos, hsml=hs, transform=aff, diffdir=diffdir),
                               oracle.Window(kind, which='ref').readout(a, pos, hsml=hs, transform=aff, diffdir=diffdir))

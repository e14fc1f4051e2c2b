"""Lookup tables of the table-driven windows (lanczos2-6, acg2-6).

The reference ships them as generated C headers (pmesh/_window_lanczos.h,
pmesh/_window_acg.h, 8192 entries each, produced by makelanczos.py / makeacg.py and
printed with 8 decimals; the step with '%e').  Here the same formulas are evaluated at
import of the first such window and rounded the same way, so the tables are equal to the
headers' constants; they are uploaded once per device (64 KB each).

    lanczos n : phi(x) = sinc(x) sinc(x/n) on [0, n), nativesupport 2n      (makelanczos.py:3-8)
    acg N     : approximated confined Gaussian of support N on [0, N/2]    (makeacg.py:4-26)
both normalised to unit integral with the trapezoid rule.
The wavelet windows (db/sym) need PyWavelets tables and are not built.
"""
import numpy

TABLE_SIZE = 8192


def _round8(phi):
    # the headers hold "%.8f" renderings of the values
    return numpy.array([float('%.8f' % v) for v in phi], dtype='f8')


def _step(x):
    return float('%e' % numpy.diff(x).mean())


def lanczos(n):
    x = numpy.linspace(0, n, TABLE_SIZE, endpoint=False)
    phi = numpy.sinc(x) * numpy.sinc(x / n)
    phi = phi / (2 * numpy.trapezoid(phi, x))
    return _round8(phi), _step(x), 2 * n


def acg(N):
    s = 1.0
    A = (N - 1) / 2.0
    x = numpy.linspace(0, N * 0.5, TABLE_SIZE, endpoint=True)
    y = x + A

    def G(y):
        return numpy.exp(-0.25 * ((y - A) / s) ** 2)
    phi = G(y) - G(-0.5) * (G(y + N) + G(y - N)) / (G(-0.5 + N) + G(-0.5 - N))
    phi = phi / (2 * numpy.trapezoid(phi, x))
    return _round8(phi), _step(x), N


_cache = {}


def table(kind):
    """(values f8[8192], step, nativesupport) for 'lanczos2'..'lanczos6', 'acg2'..'acg6'"""
    if kind not in _cache:
        if kind.startswith('lanczos'):
            _cache[kind] = lanczos(int(kind[7:]))
        elif kind.startswith('acg'):
            _cache[kind] = acg(int(kind[3:]))
        else:
            raise KeyError(kind)
    return _cache[kind]

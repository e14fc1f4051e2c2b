"""walk form vs direct kernels on small cases, one line per (window, canvas, position, mass) type"""
import sys
import numpy, torch
from pmesh_amd import backend, window
from pmesh_amd.window import Affine, windows
be = backend.get()
rs = numpy.random.RandomState(1)
shape = (64, 64, 64)
def log(*a):
    print(*a, flush=True)
for n in (20000, 400000):
    pos_h = rs.uniform(-10, 74, size=(n, 3))
    mass_h = rs.uniform(0.5, 1.5, size=n)
    aff = Affine(3, period=shape)
    for name in ('tsc', 'pcs'):
        W = windows[name]
        for cdt in (torch.float64, torch.float32):
            for pdt in ('f8', 'f4'):
                for mdt in (None, 'f8', 'f4'):
                    pos = torch.from_numpy(pos_h.astype(pdt)).to(be.device)
                    mass = None if mdt is None else torch.from_numpy(mass_h.astype(mdt)).to(be.device)
                    res = []
                    for form in ('direct', 'never', 'always'):
                        window.BINNED = 'never' if form == 'direct' else 'always'
                        window.WALK = form if form != 'direct' else 'never'
                        window.clear_bin_cache()
                        c = torch.zeros(shape, dtype=cdt, device=be.device)
                        W.paint(c, pos, mass=mass, transform=aff)
                        torch.cuda.synchronize()
                        if form == 'always' and '-v' in sys.argv: log('  paint done')
                        f = torch.from_numpy(rs.normal(size=shape)).to(cdt).to(be.device) if form == 'direct' else f
                        r = W.readout(f, pos, transform=aff)
                        torch.cuda.synchronize()
                        if form == 'always' and '-v' in sys.argv: log('  readout done')
                        res.append((c.double().cpu().numpy(), r.cpu().numpy()))
                    log('%6d %s canvas %s pos %s mass %-4s: paint tiles %.2e walk %.2e (sum %.6f / %.6f)  readout tiles %.2e walk %.2e'
                          % (n, name, str(cdt)[-7:], pdt, mdt, abs(res[1][0] - res[0][0]).max(), abs(res[2][0] - res[0][0]).max(),
                             res[2][0].sum(), res[0][0].sum(),
                             abs(res[1][1] - res[0][1]).max(), abs(res[2][1] - res[0][1]).max()))

import os, sys, numpy, torch
sys.path.insert(0, os.getcwd())
from pmesh_amd import window, backend
from pmesh_amd.window import Affine, windows
be = backend.get()
N = 64
aff = Affine(3, period=N)
rs = numpy.random.RandomState(100)
n = 20000
pos_h = rs.uniform(-40, 110, size=(n, 3))
mass_h = rs.uniform(0.5, 1.5, size=n)
name = 'tsc'
W = windows[name]
ptype = sys.argv[1]; diffdir = None if sys.argv[2] == 'none' else int(sys.argv[2])
pos_h = pos_h.astype(ptype)
pos = torch.from_numpy(pos_h).to(be.device)
pos_h = pos_h.astype('f8')
mass = torch.from_numpy(mass_h).to(be.device)
out = []
for mode in ('never', 'always'):
    window.BINNED = mode; window.WALK = 'never'; window.SORTED = 'never'
    window.clear_bin_cache()
    c = torch.zeros((N, N, N), dtype=torch.float64, device=be.device)
    W.paint(c, pos, mass=mass, transform=aff, diffdir=diffdir)
    out.append(c.cpu().numpy())
d = out[1] - out[0]
bad = numpy.argwhere(abs(d) > 1e-12)
print('bad cells', len(bad), 'sum direct', out[0].sum(), 'binned', out[1].sum(), 'mass', mass_h.sum())
base = numpy.floor(pos_h + 0.5).astype(int) - 1
tile = ((base[:, 0] % 64) // 8, (base[:, 1] % 64) // 16, (base[:, 2] % 64) // 32)
lb = numpy.stack([base[:, 0] % 8, base[:, 1] % 16, base[:, 2] % 32], axis=1)
seen = set()
for b in bad[:200]:
    # particles covering this cell
    cover = numpy.nonzero(((b[0] - base[:, 0]) % 64 < 3) & ((b[1] - base[:, 1]) % 64 < 3) & ((b[2] - base[:, 2]) % 64 < 3))[0]
    key = tuple(cover)
    if key in seen: continue
    seen.add(key)
    print('cell', tuple(b), 'diff %.4g' % d[tuple(b)], 'particles', [(int(i), tuple(int(t[i]) for t in tile), tuple(lb[i]), round(mass_h[i], 3)) for i in cover])
    if len(seen) > 12: break

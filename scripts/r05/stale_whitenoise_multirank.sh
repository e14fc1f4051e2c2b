out=gpurun_out/r05_h; mkdir -p $out
timeout 1200 python -m pytest tests/test_binned.py tests/test_whitenoise.py tests/test_multirank.py tests/test_devarr.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
python - <<'PY'
import time, torch
from pmesh_amd.pm import ParticleMesh
for N in (256, 512, 1024):
    pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
    pm.generate_whitenoise(seed=7, type='complex'); torch.cuda.synchronize()
    t0 = time.perf_counter(); pm.generate_whitenoise(seed=8, type='complex'); torch.cuda.synchronize()
    print('whitenoise %d^3 (master stream on the device): %.1f ms' % (N, 1e3 * (time.perf_counter() - t0)))
PY

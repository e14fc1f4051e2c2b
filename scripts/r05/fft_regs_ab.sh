#!/bin/bash
# register first / last passes of the column FFT kernels (stockham_pass_p2: IO) against the LDS forms, same box:
#   (here) bash scripts/build_variant.sh colregs0 "-DPMX_COL_REGS=0" pmx_colfft.hip
#   (here) bash scripts/build_variant.sh allregs0 "-DPMX_COL_REGS=0 -DPMX_ROUND_REGS=0" pmx_colfft.hip
#   (box)  bash scripts/r05/fft_regs_ab.sh
run() { PMESH_AMD_LIBRARY=$1 timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "${@:3}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages_ms']
print('%-10s %-28s %8.3f ms  r2c %.3f c2r %.3f' % ('$2', '${*:3}', d['ms_per_step'], st['r2c'], st['c2r']))"; }
for rep in 1 2; do
for args in "" "--dtype f4" "--mesh 256" "--mesh 1024 --steps 5"; do
  run "" product $args
  run $PWD/pmesh_amd/libpmesh_amd_colregs0.so colregs0 $args
  run $PWD/pmesh_amd/libpmesh_amd_allregs0.so allregs0 $args
done; done

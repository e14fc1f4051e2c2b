#!/bin/bash
# the 768-thread PCS readouts on double canvases held to 80 VGPRs (two workgroups per CU) against the unbounded build:
#   (here) bash scripts/build_variant.sh ro768w1 "-DPMX_READOUT768_WAVES=1" pmx_binned.hip
#   (box)  bash scripts/r05/readout768_ab.sh
# measured: shuffled rows (tile-ordered copy) readout 7.02 -> 6.16 ms; rows in lattice order (lean form, whole mesh) unchanged
run() { PMESH_AMD_LIBRARY=$1 timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 3 "${@:3}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages_ms']
print('%-10s %-40s %8.3f ms  bin %.3f paint %.3f readout %.3f' % ('$2', '${*:3}', d['ms_per_step'], st['bin'], st['paint'], st['readout']))"; }
for rep in 1 2; do
for args in "--window pcs --data shuffled" "--window pcs"; do
  run "" product $args
  run $PWD/pmesh_amd/libpmesh_amd_ro768w1.so unbounded $args
done; done

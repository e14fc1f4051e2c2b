#!/bin/bash
# what the kernel forms for blocks of ANY shape (slab / pencil ranks) cost against the whole-mesh forms, on one rank's
# whole mesh where both apply:
#   (here) bash scripts/build_variant.sh general "-DPMX_GENERAL_FORMS_ONLY=1" pmx_binned.hip
#   (box)  bash scripts/r05/general_forms_ab.sh
run() { PMESH_AMD_LIBRARY=$1 timeout 600 python bench.py --no-cpu-baseline --steps 15 --warmup 4 "${@:3}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages_ms']
print('%-10s %-40s %8.3f ms  bin %.3f paint %.3f readout %.3f' % ('$2', '${*:3}', d['ms_per_step'], st['bin'], st['paint'], st['readout']))"; }
for rep in 1 2; do
for args in "" "--window tsc --dtype f4 --gradient 0" "--window pcs" "--window tsc" "--window pcs --dtype f4"; do
  run "" product $args
  run $PWD/pmesh_amd/libpmesh_amd_general.so general $args
done; done

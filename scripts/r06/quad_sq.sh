#!/bin/bash
# SQ counters of the PCS paint kernel, four lanes per particle (PMX_QUAD_MIN=0: every tile), the product's per-tile choice, one lane
# per particle (2^30): LDS bank conflicts and LDS waits per wave cycle (scripts/sq_profile.sh) -> gpurun_out/<tag>_*/sq_summary.txt
tag=${1:-r06_quadsq}
for qm in 0 auto 1073741824; do
  if [ $qm = auto ]; then unset PMX_QUAD_MIN; else export PMX_QUAD_MIN=$qm; fi
  echo "== PMX_QUAD_MIN=$qm: 256^3 Zel'dovich set, PCS f8"
  scripts/sq_profile.sh ${tag}_256_$qm --mesh 256 --window pcs --data clustered | grep "paint_tile_kernel"
  echo "== PMX_QUAD_MIN=$qm: config 5's per-GPU load"
  scripts/sq_profile.sh ${tag}_c5_$qm --mesh 1024 --window pcs --data clustered --double 1 --mass array | grep "paint_tile_kernel"
done

#!/bin/bash
# A/B of the four-lanes-per-particle PCS deposit (PMX_QUAD_PCS, csrc/pmx_binned.hip) against the lane-per-particle loop
# on one box: scripts/build_variant.sh noquad "-DPMX_QUAD_PCS=0" first (here: the library travels with the snapshot).
out=gpurun_out/${1:-r06_quad}; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-64s %8.3f ms  bin %.2f paint %.3f r2c %.2f c2r %.2f readout %.3f" % (sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
PY
}
for rep in 1 2; do
for cfg in "--window pcs" "--window pcs --data clustered" "--window pcs --data shuffled" "--window pcs --drift 4" "--window pcs --mass array" "--window pcs --gradient 0" "--mesh 256 --window pcs --data clustered"; do
  for lib in quad noquad; do
    if [ $lib = noquad ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_noquad.so; else unset PMESH_AMD_LIBRARY; fi
    timeout 300 python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 > $out/r.json 2> $out/r.err && line $out/r.json "[$lib] $cfg"
  done
done
done
for lib in quad noquad; do
  if [ $lib = noquad ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_noquad.so; else unset PMESH_AMD_LIBRARY; fi
  timeout 600 python bench.py --mesh 1024 --window pcs --data clustered --double 1 --mass array --no-cpu-baseline --steps 4 --warmup 2 > $out/r.json 2> $out/r.err && line $out/r.json "[$lib] C5 per-GPU load (1024^3, 2x1024^3 clustered, masses, PCS f8)"
done

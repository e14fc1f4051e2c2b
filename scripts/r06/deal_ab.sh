#!/bin/bash
# the crowded-tile deal of the one-lane deposit loops (PMX_DEAL_CROWDED) against a build without it, same box
# (scripts/build_variant.sh nodeal "-DPMX_DEAL_CROWDED=0")
out=gpurun_out/${1:-r06_deal}; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-44s %8.3f ms  paint %.3f" % (sys.argv[2], d["ms_per_step"], st["paint"]))
PY
}
for rep in 1 2 3; do
for cfg in "" "--dtype f4" "--mesh 256" "--data clustered" "--mesh 1024 --steps 5 --warmup 2"; do
  for lib in product nodeal; do
    if [ $lib = product ]; then unset PMESH_AMD_LIBRARY; else export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so; fi
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && line $out/r.json "[$lib] $cfg"
  done
done
done

#!/bin/bash
# scripts/r06/lib_ab.sh <variant> <tag>: the product library against pmesh_amd/libpmesh_amd_<variant>.so (scripts/build_variant.sh),
# alternating, twice, over the configurations of the sweep that exercise the tile kernels; stage times from bench.py
v=$1; out=gpurun_out/${2:-r06_libab}; mkdir -p $out
for rep in 1 2; do
for cfg in "" "--config c3" "--window tsc" "--window pcs" "--data clustered" "--drift 4" "--mesh 256" "--mesh 1024 --steps 5 --warmup 2" "--dtype f4"; do
  for lib in product $v; do
    if [ $lib = product ]; then unset PMESH_AMD_LIBRARY; else export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$v.so; fi
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && python - $out/r.json "[$lib] $cfg" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-50s %8.3f ms  bin %.3f paint %.3f r2c %.3f c2r %.3f readout %.3f" % (sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
PY
  done
done
done

#!/bin/bash
# scripts/r06/lib_abn.sh <tag> <lib> [<lib> ...]: several variant libraries (pmesh_amd/libpmesh_amd_<lib>.so), in turn, twice,
# over the configurations that exercise the tile kernels
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
for cfg in "" "--config c3" "--window tsc" "--data clustered" "--drift 4" "--mesh 1024 --steps 5 --warmup 2" "--dtype f4" "--mesh 256"; do
  for lib in "$@"; do
    export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && python - $out/r.json "[$lib] $cfg" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-50s %8.3f ms  bin %.3f paint %.3f r2c %.3f c2r %.3f readout %.3f" % (sys.argv[2][:50], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
PY
  done
done
done

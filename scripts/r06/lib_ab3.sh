#!/bin/bash
# scripts/r06/lib_ab3.sh <libA> <libB> <tag>: two libraries by name (pmesh_amd/libpmesh_amd_<name>.so), alternating, twice,
# over the configurations whose transforms differ in length and precision
a=$1; b=$2; out=gpurun_out/${3:-r06_libab}; mkdir -p $out
for rep in 1 2; do
for cfg in "" "--config c3" "--mesh 1024 --steps 5 --warmup 2" "--mesh 768 --steps 5 --warmup 2" "--mesh 384" "--mesh 256" "--mesh 1024 --dtype f4 --steps 5 --warmup 2" "--mesh 640 --steps 5 --warmup 2"; do
  for lib in $a $b; do
    export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && python - $out/r.json "[$lib] $cfg" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-60s %8.3f ms  r2c %.3f c2r %.3f" % (sys.argv[2][:60], d["ms_per_step"], st["r2c"], st["c2r"]))
PY
  done
done
done

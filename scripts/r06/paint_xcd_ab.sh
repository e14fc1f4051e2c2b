#!/bin/bash
out=gpurun_out/r06_px; mkdir -p $out
for rep in 1 2; do
for cfg in "" "--data clustered" "--window pcs --data clustered" "--drift 4" "--config c3" "--mesh 1024 --double 1 --mass array --window pcs --data clustered --steps 3 --warmup 1"; do
  for lib in px0 px1; do
    export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && python - $out/r.json "[$lib] $cfg" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-50s %8.3f ms  bin %.3f paint %.3f r2c %.3f c2r %.3f readout %.3f" % (sys.argv[2][:50], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
PY
  done
done
done

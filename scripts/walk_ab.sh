#!/bin/bash
# A/B of the two binned forms on the GPU box: tile kernels (PMX walk 'never') vs walk kernels
out=gpurun_out/walk_ab; mkdir -p $out
for cfg in "--window tsc" "--window pcs" "--window tsc --dtype f4 --gradient 0" "--window pcs --data clustered" "--window tsc --data clustered" "--window cic" "$@"; do
  for walk in never always; do
    PMESH_AMD_WALK=$walk timeout 300 python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 > $out/r.json 2>$out/r.err || tail -3 $out/r.err
    python - "$walk" "$cfg" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/walk_ab/r.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[walk=%s] %-45s %.3f ms  bin %.2f paint %.2f readout %.2f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["readout"]))
except Exception as e:
    print("[walk=%s] %s FAILED %r" % (sys.argv[1], sys.argv[2], e))
PY
  done
done

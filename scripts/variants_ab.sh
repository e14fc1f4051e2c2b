#!/bin/bash
# scripts/variants_ab.sh "<variant names ('' = the product library)>" "<bench args>" ...: stage times per variant
vars=$1; shift
for cfg in "$@"; do
  for v in $vars; do
    lib=$PWD/pmesh_amd/libpmesh_amd_$v.so; [ "$v" = "base" ] && lib=$PWD/pmesh_amd/libpmesh_amd.so
    PMESH_AMD_LIBRARY=$lib timeout 300 python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/v.json 2>gpurun_out/v.err
    python - "$v" "$cfg" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/v.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[%-6s] %-42s %.3f ms  bin %.2f paint %.2f r2c %.3f c2r %.3f readout %.2f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
except Exception as ex:
    print("[%s] %s FAILED" % (sys.argv[1], sys.argv[2]), open("gpurun_out/v.err").read()[-300:])
PY
  done
done

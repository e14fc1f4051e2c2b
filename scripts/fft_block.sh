#!/bin/bash
for mb in ${MBS:-0 32 64 96 128 192}; do
  PMESH_AMD_L3_BLOCK_MB=$mb timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" > gpurun_out/v.json 2>gpurun_out/v.err
  python - $mb <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/v.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("block %4s MB: %.3f ms  paint %.2f r2c %.3f c2r %.3f readout %.2f" % (sys.argv[1], d["ms_per_step"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
except Exception as ex:
    print("FAILED", open("gpurun_out/v.err").read()[-300:])
PY
done

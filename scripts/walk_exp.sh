#!/bin/bash
for e in "" _exp1 _exp3 _exp4 _exp5 _exp6; do
  PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd$e.so PMESH_AMD_BENCH_NOCHECK=1 PMESH_AMD_WALK=always timeout 300 python bench.py --window ${1:-tsc} --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/exp.json 2>gpurun_out/exp.err
  python - "$e" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/exp.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("EXP %s: paint %.2f readout %.2f" % (sys.argv[1], st["paint"], st["readout"]))
except Exception as ex:
    print("EXP %s failed" % sys.argv[1], open("gpurun_out/exp.err").read()[-300:])
PY
done

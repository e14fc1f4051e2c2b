#!/bin/bash
# which tiles take the four-lanes-per-particle PCS deposit: sweep of PMX_QUAD_MIN (entries per tile from which on)
out=gpurun_out/${1:-r06_quadmin}; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-72s %8.3f ms  paint %.3f" % (sys.argv[2], d["ms_per_step"], st["paint"]))
PY
}
for cfg in "--window pcs" "--window pcs --data clustered" "--window pcs --drift 4" "--window pcs --drift 8" "--mesh 256 --window pcs --data clustered" "--window pcs --particles 640" "--mesh 1024 --window pcs --data clustered --double 1 --mass array --steps 4 --warmup 2"; do
  for qm in 0 4608 5120 6144 8192 1073741824; do
    PMX_QUAD_MIN=$qm timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && line $out/r.json "[quad_min $qm] $cfg"
  done
done

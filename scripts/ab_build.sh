#!/bin/bash
# A/B of a compile-time switch on the GPU box: scripts/ab_build.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# rebuilds pmx_binned.hip with each flag set and prints the stage times of CIC/TSC/PCS f8, config 3
# and the clustered set.
out=gpurun_out/ab; mkdir -p $out
i=0
for flags in "$@"; do
  i=$((i+1))
  (cd pmesh_amd/csrc && touch pmx_binned.hip && make EXTRA="$flags" 2>&1 | grep -E "error|warning: v" | head -3)
  for cfg in "--window cic" "--window tsc" "--window pcs" "--window tsc --dtype f4 --gradient 0" "--dtype f4" "--data clustered" "--mesh 256"; do
    timeout 300 python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 > $out/r.json 2>$out/r.err
    python - "$flags" "$cfg" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/ab/r.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
print("[%s] %-40s %.3f ms  bin %.2f paint %.2f readout %.2f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["readout"]))
PY
  done
done

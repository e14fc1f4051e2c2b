#!/bin/bash
# the readout's z segments with the carried face (PMX_READOUT_ZWALK, csrc/pmx_binned.hip) against one tile per workgroup
# (PMX_READOUT_RSEG=1) and fixed segment lengths, same box
out=gpurun_out/${1:-r06_zwalk}; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-60s %8.3f ms  readout %.3f" % (sys.argv[2], d["ms_per_step"], st["readout"]))
PY
}
for cfg in "" "--config c3" "--window tsc" "--window pcs" "--data clustered" "--dtype f4" "--mesh 256" "--mesh 1024 --steps 5 --warmup 2" "--drift 4"; do
  for rs in auto 1 2 4 8 16; do
    if [ $rs = auto ]; then unset PMX_READOUT_RSEG; else export PMX_READOUT_RSEG=$rs; fi
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && line $out/r.json "[rseg $rs] $cfg"
  done
done

#!/bin/bash
# the measured table of DESIGN.md: one bench.py line per configuration -> gpurun_out/sweep/*.json
out=gpurun_out/sweep; mkdir -p $out
i=0
while read -r name args; do
  [ -z "$name" ] && continue
  timeout 900 python bench.py $args --no-cpu-baseline > $out/$name.json 2> $out/$name.err
  python - "$name" <<'PY'
import json, sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/sweep/%s.json" % n).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-14s %8.3f ms %.3e p/s  bin %.2f paint %.2f r2c %.2f c2r %.2f readout %.2f" % (n, d["ms_per_step"], d["value"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
PY
done <<'CFG'
headline --mesh 512
clustered --mesh 512 --data clustered
c2_256 --mesh 256
c3_tsc_f4 --mesh 512 --window tsc --dtype f4 --gradient 0
tsc_f8 --mesh 512 --window tsc
pcs_f8 --mesh 512 --window pcs
cic_f4 --mesh 512 --dtype f4
m384 --mesh 384
m768 --mesh 768
m1024 --mesh 1024 --steps 5
c5_shard --mesh 1024 --double 1 --mass array --window pcs --data clustered --steps 5
CFG

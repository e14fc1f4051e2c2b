#!/bin/bash
# where the tile-ordered copy of the positions starts to pay: the headline with rows less and less in lattice order
# (N(0, drift) cells per step on a lattice; fully shuffled), the plan's form forced both ways and left to itself
# bash scripts/r05/order_threshold.sh
run() { PMESH_AMD_SORTED=$1 timeout 600 python bench.py --no-cpu-baseline --steps 9 --warmup 3 "${@:3}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages_ms']
print('%-10s %-7s %8.3f ms  bin %.2f paint %.2f readout %.2f  overflows %s sorted_plans %s' % ('$2', '$1', d['ms_per_step'], st['bin'], st['paint'], st['readout'], d.get('bin_overflows'), d.get('sorted_plans')))"; }
for d in 1.0 2.0 4.0 8.0; do for s in never always auto; do run $s drift$d --drift $d; done; done
for s in never always auto; do run $s shuffled --data shuffled; done

#!/bin/bash
# one bench.py line per configuration of the measured table in DESIGN.md -> gpurun_out/<tag>/*.json
tag=${1:-sweep}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = d['stages_ms']
    print('%-22s %8.3f ms %.3e p/s  bin %.2f paint %.2f r2c %.2f c2r %.2f readout %.2f  frac %.3f with_bin %.3f ovf %d' % (
        sys.argv[2], d['ms_per_step'], d['value'], st['bin'], st['paint'], st['r2c'], st['c2r'], st['readout'],
        d['roofline']['frac'], d['roofline']['with_bin_frac'], d['bin_overflows']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run headline
run headline_nodrift --drift 0
run headline_drift05 --drift 0.5
run headline_drift1 --drift 1.0
run headline_drift2 --drift 2.0
run headline_drift4 --drift 4.0
run headline_drift8 --drift 8.0
run headline_cold_plan --cold-plan 1 --steps 5
run headline_count_jitter --count-jitter 0.01
run headline_pos_columns6 --pos-columns 6
run clustered --data clustered
run config2_256 --mesh 256
run config3_tsc_f4_grad --window tsc --dtype f4 --gradient 0
run tsc_f8 --window tsc
run pcs_f8 --window pcs
run cic_f4 --dtype f4
run m384 --mesh 384
run m768 --mesh 768 --steps 5
run m1024 --mesh 1024 --steps 5
run c5_shard --mesh 1024 --double 1 --mass array --window pcs --data clustered --steps 3 --warmup 1
run shuffled --data shuffled --steps 5
run host_arrays --host-arrays 1 --steps 5

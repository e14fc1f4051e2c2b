#!/bin/bash
# the FFT-sensitive lines of scripts/sweep.sh
tag=${1:-sweep_short}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = d['stages_ms']
    print('%-22s %8.3f ms %.3e p/s  bin %.2f paint %.2f r2c %.2f c2r %.2f readout %.2f' % (
        sys.argv[2], d['ms_per_step'], d['value'], st['bin'], st['paint'], st['r2c'], st['c2r'], st['readout']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run headline
run config2_256 --mesh 256
run config3_tsc_f4_grad --window tsc --dtype f4 --gradient 0
run cic_f4 --dtype f4
run m384 --mesh 384
run m640 --mesh 640 --steps 5
run m768 --mesh 768 --steps 5
run m1024 --mesh 1024 --steps 5

#!/bin/bash
# per-kernel times of a caller's time step as examples/nbody.py writes it (scripts/nbody_steps.py): bash scripts/r05/nbody_kstats.sh
repo=$PWD; out=$PWD/gpurun_out/r05_nbody; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $repo/scripts/nbody_steps.py 512 5 > $out/log 2>&1
cd $repo
tail -4 $out/log
python3 - $(find $out/stats -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print('%-96s calls %4s avg %8.1f us  total %8.3f ms' % (r['Name'].replace('void pmx::', '').replace('void at::native::', 'at::')[:96], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $out/stats

#!/bin/bash
# one rank with / without a layout: per-kernel times (is any work done twice?)
repo=$PWD
for l in 0 1; do
out=$repo/gpurun_out/la$l; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $repo/scripts/layout_anomaly.py $l > $out/log 2>&1)
grep "per cycle" $out/log || tail -5 $out/log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-100s calls %5s total %9.3f ms avg %9.1f us' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
rm -rf $out
done

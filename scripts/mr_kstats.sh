#!/bin/bash
# per-kernel time of the P-rank cycle with all ranks as threads on one GPU: scripts/mr_kstats.sh <tag> [mr_probe args]
tag=$1; shift
repo=$PWD; out=$PWD/gpurun_out/$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $repo/scripts/mr_probe.py "$@" > $out/log 2>&1
rc=$?
cd $repo
# a run that did not finish (out of memory, a failed check) leaves NO kernel table: a table divided by the cycles it
# was asked for would pass for a measurement (round 3's "367 ms" came from a run that died in its third cycle)
if [ $rc -ne 0 ] || ! grep -q "ms wall per cycle" $out/log; then
  echo "mr_kstats.sh: the probed program FAILED (exit $rc): no kernel table written" >&2
  tail -5 $out/log >&2
  rm -rf $out/stats gpurun_out/${tag}_kernel_stats.csv
  exit 1
fi
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
tail -2 $out/log
python3 - gpurun_out/${tag}_kernel_stats.csv "$@" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
args = sys.argv[2:]
steps = int(args[args.index('--steps') + 1]) if '--steps' in args else 5
warm = int(args[args.index('--warmup') + 1]) if '--warmup' in args else 2
tot = sum(float(r['TotalDurationNs']) for r in rows if 'pmx::' in r['Name'] and 'synth' not in r['Name'])
print('pmx kernels: %.3f ms per cycle summed over all ranks (%d cycles incl. warm-up)' % (tot / 1e6 / (steps + warm), steps + warm))
for r in rows[:24]:
    print('%-100s calls %5s total %9.3f ms avg %9.1f us' % (r['Name'].replace('void pmx::', '')[:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
rm -rf $out/stats

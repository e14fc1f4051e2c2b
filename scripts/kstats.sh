#!/bin/bash
# scripts/kstats.sh <tag> [bench args]: per-kernel times of a few cycles -> gpurun_out/<tag>_kernel_stats.csv (+ top lines on stdout)
tag=$1; shift
repo=$PWD; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $out/stats.log 2>&1
cd $repo
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
cp $(find $out/stats -name "*kernel_trace.csv" | head -1) gpurun_out/${tag}_kernel_trace.csv
rm -rf $out
python3 - gpurun_out/${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print('%-110s calls %5s total %9.3f ms avg %9.1f us' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
python3 - gpurun_out/${tag}_kernel_trace.csv "${KSHOW:-movers_kernel}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        print('%-60s %9.1f us  grid %s wg %s vgpr %s sgpr %s scratch %s lds %s' % (r['Kernel_Name'][:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
              r.get('Grid_Size_X'), r.get('Workgroup_Size_X'), r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('Scratch_Size'), r.get('LDS_Block_Size')))
PY
rm -f gpurun_out/${tag}_kernel_trace.csv

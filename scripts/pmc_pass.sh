#!/bin/bash
# one PMC pass over a bench run, raw totals per kernel: scripts/pmc_pass.sh <tag> "<counters>" [bench args...]
tag=$1; counters=$2; shift 2
out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/sq -o p -- python3 $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/sq.log 2>&1
cd $repo
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/sq/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'].split('(')[0][:60]
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    n[(k, row['Counter_Name'])] += 1
dur = collections.defaultdict(list)
for g in glob.glob(out + '/sq/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(g)):
        dur[row['Kernel_Name'].split('(')[0][:60]].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
with open(out + '/pmc_summary.txt', 'a') as o:
    for k, c in acc.items():
        if 'walk' not in k and 'tile_kernel' not in k: continue
        launches = max(n[(k, name)] for name in c)
        d = sum(dur.get(k, [0])) / max(1, len(dur.get(k, [0])))
        line = '%-50s launches %d avg %.0f us | per launch: ' % (k, launches, d / 1e3) + ' '.join('%s=%.4g' % (name, c[name] / launches) for name in sorted(c))
        print(line); o.write(line + '\n')
PY
rm -rf $out/sq

#!/bin/bash
# SQ counters of the tile kernels (one pass, 8 SQ slots): where do the waves spend their cycles?
#   scripts/sq_profile.sh <tag> [bench args...]  -> gpurun_out/<tag>/sq_<kernel>.txt
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
   --output-format csv -d $out/sq -o p -- python3 $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/sq.log 2>&1
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
with open(out + '/sq_summary.txt', 'w') as o:
    for k, c in acc.items():
        if 'tile' not in k and 'halo' not in k and 'bin_' not in k and 'fft' not in k: continue
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        line = '%-62s ' % k + ' '.join('%s=%.3f' % (name.replace('SQ_', ''), c[name] / wc) for name in sorted(c) if name != 'SQ_WAVE_CYCLES')
        print(line); o.write(line + '\n')
PY
rm -rf $out/sq

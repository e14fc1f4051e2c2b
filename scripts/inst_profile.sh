#!/bin/bash
# dynamic instruction counts of the particle kernels (one SQ pass)
#   scripts/inst_profile.sh <tag> [bench args...]  -> gpurun_out/<tag>/inst_summary.txt
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
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
with open(out + '/inst_summary.txt', 'w') as o:
    for k, c in acc.items():
        if 'tile' not in k and 'bin_' not in k and 'fft' not in k: continue
        launches = n[(k, 'SQ_WAVES')]
        w = c.get('SQ_WAVES', 0) or 1
        line = '%-58s launches %d waves/launch %.0f  per wave: ' % (k, launches, w / launches) + ' '.join(
            '%s=%.0f' % (name.replace('SQ_', ''), c[name] / w) for name in sorted(c) if name != 'SQ_WAVES')
        print(line); o.write(line + '\n')
PY
rm -rf $out/sq

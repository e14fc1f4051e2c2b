#!/bin/bash
# host time of the HIP runtime calls of the one-rank cycle (device idle: 128^3), and of hipLaunchKernel per kernel
# (rocprofv3 --hip-trace --kernel-trace, joined on the correlation id): bash scripts/r05/host_api_stats.sh
repo=$PWD; out=$PWD/gpurun_out/r05_host_api; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $repo/scripts/host_profile0.py ${1:-128} > $out/log 2>&1
cd $repo
python3 - $(find $out/stats -name "*hip_api_stats.csv" | head -1) $(find $out/stats -name "*hip_api_trace.csv" | head -1) $(find $out/stats -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print('%-40s calls %7s avg %8.2f us total %9.3f ms' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
api = {}
for r in csv.DictReader(open(sys.argv[2])):
    if r['Function'] == 'hipLaunchKernel':
        api[r['Correlation_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
per = {}
for r in csv.DictReader(open(sys.argv[3])):
    d = api.get(r['Correlation_Id'])
    if d is None: continue
    e = per.setdefault(r['Kernel_Name'].split('(')[0][:70], [0, 0, []]); e[0] += 1; e[1] += d; e[2].append(d)
for k, (n, t, ds) in sorted(per.items(), key=lambda kv: -kv[1][1])[:22]:
    ds.sort()
    print('%-72s launches %5d host avg %7.2f us median %7.2f us' % (k, n, t / n / 1e3, ds[len(ds) // 2] / 1e3))
PY
rm -rf $out/stats

# per-kernel times of one rank's slab work alone on the GPU (scripts/slab_rank_probe.py): bash scripts/r05/slab_rank_kstats.sh 8 cic
P=${1:-8}; W=${2:-cic}
repo=$PWD; out=$PWD/gpurun_out/r05_slab_${P}_$W; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $repo/scripts/slab_rank_probe.py $P $W > $out/log 2>&1
cd $repo
grep rank $out/log
python3 - $(find $out/stats -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    if 'synth' in r['Name'] or 'at::' in r['Name']: continue
    print('%-86s calls %4s avg %8.1f us  total %8.3f ms' % (r['Name'].replace('void pmx::', '').replace('pmx::', '')[:86], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $out/stats

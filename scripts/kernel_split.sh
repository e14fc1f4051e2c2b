#!/bin/bash
# paint_tile_kernel alone, per variant library: scripts/kernel_split.sh "<variants>" "<bench args>"...
vars=$1; shift
for cfg in "$@"; do
  for v in $vars; do
    lib=$PWD/pmesh_amd/libpmesh_amd_$v.so; [ "$v" = "base" ] && lib=$PWD/pmesh_amd/libpmesh_amd.so
    repo=$PWD; out=$PWD/gpurun_out/ks_$v; rm -rf $out; mkdir -p $out
    (cd /tmp && export TMPDIR=/tmp && PMESH_AMD_BENCH_NOCHECK=1 PMESH_AMD_LIBRARY=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $repo/bench.py $cfg --no-cpu-baseline --steps 6 --warmup 2 > $out/log 2>&1)
    python3 - $out "$v" "$cfg" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    n = r['Name']
    if 'paint_tile' in n or 'halo_merge' in n or 'readout_tile' in n or 'bin_block' in n:
        print('[%-8s] %-40s %-22s avg %8.1f us' % (sys.argv[2], sys.argv[3], n.split('(')[0].replace('void pmx::', '')[:22], float(r['AverageNs']) / 1e3))
PY
    rm -rf $out
  done
done

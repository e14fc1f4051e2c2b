#!/bin/bash
# HBM bytes of one column pass over dense (257-mode) and padded (264) lines, tiles in blockIdx order (variant xcd0) and in
# XCD order (the product): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, folded by scripts/pmc_summary.py
repo=$PWD; out=$PWD/gpurun_out/${1:-r06_colxcd_pmc}; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in xcd0 product; do
  if [ $lib = xcd0 ]; then export PMESH_AMD_LIBRARY=$repo/pmesh_amd/libpmesh_amd_xcd0.so; else unset PMESH_AMD_LIBRARY; fi
  for B in 257 264; do
    d=$out/${lib}_B$B
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/fetch -o p -- python3 $repo/scripts/r06/col_one.py 512 $B > $d.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/write -o p -- python3 $repo/scripts/r06/col_one.py 512 $B >> $d.log 2>&1
    (cd $repo && python3 scripts/pmc_summary.py $d/fetch $d/write $d.csv > /dev/null 2>&1)
    echo "== $lib B=$B"; grep "algorithmic" $d.log | head -1; grep colfft $d.csv | cut -c1-60,140-
    rm -rf $d
  done
done

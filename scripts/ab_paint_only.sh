#!/bin/bash
# timing experiments on the paint tile kernel (PMX_EXPERIMENT_LDS variants give WRONG results on
# purpose: they only show where the time goes): prints the rocprof average of paint_tile_kernel
for flags in "$@"; do
  (cd pmesh_amd/csrc && make EXTRA="$flags" 2>&1 | grep -E " error" | head -3)
  for w in tsc pcs cic; do
    d=$PWD/gpurun_out/exp_$w; rm -rf $d; mkdir -p $d
    (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $OLDPWD/scripts/paint_only.py $w > /dev/null 2>&1)
    echo "[$flags] $w $(grep paint_tile $d/p_kernel_stats.csv | awk -F'",' '{split($2,a,","); print a[3]/1000 " us"}')"
  done
done

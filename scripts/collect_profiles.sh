#!/bin/bash
# scripts/collect_profiles.sh <tag>: what scripts/round_profiles.sh <tag> left under gpurun_out/ -> profiles/<tag>_* (the
# files DESIGN.md / profiles/README.md quote), and profiles/traffic.json regenerated from them
tag=$1
for n in headline c3 tsc pcs pcs_clustered c5shard; do
  [ -d gpurun_out/${tag}_$n ] || continue
  mkdir -p profiles/${tag}_$n
  for f in bench.json kernel_stats.csv pmc_hbm.csv; do cp gpurun_out/${tag}_$n/$f profiles/${tag}_$n/ 2>/dev/null; done
done
mkdir -p profiles/${tag}_sweep; cp gpurun_out/${tag}_sweep/*.json profiles/${tag}_sweep/
cp gpurun_out/${tag}_inst_counts.txt profiles/ 2>/dev/null
cp gpurun_out/${tag}_gputest.log profiles/ 2>/dev/null
for m in 512:512 1024_c4:c4_slab_1024 1024_c5:c5_pencil_1024 1024_c5_nomigrate:c5_pencil_1024_nomigrate; do
  a=${m%%:*}; b=${m#*:}
  [ -f gpurun_out/${tag}_mr8_${a}_kernel_stats.csv ] && cp gpurun_out/${tag}_mr8_${a}_kernel_stats.csv profiles/${tag}_multirank8_${b}_on_one_gpu_kernel_stats.csv
  [ -f gpurun_out/${tag}_mr8_${a}/log ] && grep -v "^[EWI]2026\|rocprofv3" gpurun_out/${tag}_mr8_${a}/log > profiles/${tag}_multirank8_${b}.log
done
python scripts/make_traffic.py $tag profiles/${tag}_* 

#!/bin/bash
# Everything a round's DESIGN.md quotes, in one call on the GPU box:  scripts/round_profiles.sh r04_a
#   the -m gpu test log, the bench line, profile_all.sh (stats + FETCH + WRITE for the headline and the weak
#   configurations), the instruction-count pass, the sweep, and the kernel tables of the 8-rank cycles
#   (thread ranks on the one GPU: 512^3 slabs, 1024^3 slabs = config 4, 1024^3 2 x 4 pencils PCS = config 5's decomposition)
tag=$1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -rs > gpurun_out/${tag}_gputest.log 2>&1; tail -4 gpurun_out/${tag}_gputest.log
scripts/profile_all.sh $tag > gpurun_out/${tag}_profile_all.log 2>&1; grep -c kernel gpurun_out/${tag}_profile_all.log
for cfg in "headline:" "c3:--window tsc --dtype f4 --gradient 0" "pcs:--window pcs" "tsc:--window tsc"; do
  n=${cfg%%:*}; a=${cfg#*:}
  scripts/inst_profile.sh ${tag}_inst_$n $a > /dev/null 2>&1
  echo "== $n ($a)" >> gpurun_out/${tag}_inst_counts.txt; cat gpurun_out/${tag}_inst_$n/inst_summary.txt >> gpurun_out/${tag}_inst_counts.txt
done
cat gpurun_out/${tag}_inst_counts.txt | cut -c1-220
scripts/sweep.sh ${tag}_sweep
scripts/mr_kstats.sh ${tag}_mr8_512 --ranks 8 --mesh 512 --steps 5 | tail -3
scripts/mr_kstats.sh ${tag}_mr8_1024_c4 --ranks 8 --mesh 1024 --steps 2 --warmup 1 | tail -3
scripts/mr_kstats.sh ${tag}_mr8_1024_c5 --ranks 8 --np 2x4 --mesh 1024 --window pcs --data clustered --double 1 --mass array --pos-dtype f4 --steps 2 --warmup 1 --migrate 1 | tail -3
scripts/mr_kstats.sh ${tag}_mr8_1024_c5_nomigrate --ranks 8 --np 2x4 --mesh 1024 --window pcs --data clustered --double 1 --mass array --pos-dtype f4 --steps 2 --warmup 1 | tail -3
grep -h "peak device\|wall per cycle" gpurun_out/${tag}_mr8_*/log

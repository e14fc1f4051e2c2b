# the profile set of round 5 (run on the GPU box): stats + FETCH + WRITE passes per configuration, the sweep, the 8-rank tables
tag=${1:-r05_p}
WITH_C5=1 scripts/profile_all.sh $tag > gpurun_out/${tag}_profile_all.log 2>&1; grep -c kernel gpurun_out/${tag}_profile_all.log
scripts/sweep.sh ${tag}_sweep
scripts/mr_kstats.sh ${tag}_mr8_512 --ranks 8 --mesh 512 --steps 12 --warmup 2 | tail -3
scripts/mr_kstats.sh ${tag}_mr8_1024_c4 --ranks 8 --mesh 1024 --steps 10 --warmup 2 | tail -3
# (the one-time kernels of decompose and of the first plan build are in these sums: 12-14 cycles keep them under 3 %)

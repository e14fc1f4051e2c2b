#!/bin/bash
# The profile set of a round for the headline AND the configurations whose kernels are the weak ones
# (run on the GPU box from the repository root):   scripts/profile_all.sh r03_a
# -> gpurun_out/<tag>_<name>/{bench.json, kernel_stats.csv, pmc_hbm.csv}; copy them to profiles/ and run
#    python scripts/make_traffic.py <tag> profiles/<tag>_*   to regenerate profiles/traffic.json
tag=$1
scripts/profile_round.sh ${tag}_headline
scripts/profile_round.sh ${tag}_c3 --no-cpu-baseline --window tsc --dtype f4 --gradient 0
scripts/profile_round.sh ${tag}_pcs --no-cpu-baseline --window pcs
scripts/profile_round.sh ${tag}_pcs_clustered --no-cpu-baseline --window pcs --data clustered
scripts/profile_round.sh ${tag}_tsc --no-cpu-baseline --window tsc
if [ "${WITH_C5:-1}" = "1" ]; then
  scripts/profile_round.sh ${tag}_c5shard --no-cpu-baseline --mesh 1024 --double 1 --mass array --window pcs --data clustered --steps 3 --warmup 1
fi
for d in gpurun_out/${tag}_*; do echo "== $d"; tail -c 900 $d/bench.json; echo; head -8 $d/kernel_stats.csv | cut -c1-200; done

#!/bin/bash
# The profile set of a round (run on the GPU box from the repository root):
#   scripts/profile_round.sh <tag> [bench args...]
# -> gpurun_out/<tag>/{kernel_stats.csv, pmc_hbm.csv, bench.json}
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
repo=$PWD
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $repo/bench.py --no-cpu-baseline "$@" --steps 5 --warmup 2 > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $repo/bench.py --no-cpu-baseline "$@" --steps 2 --warmup 1 > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 $repo/bench.py --no-cpu-baseline "$@" --steps 2 --warmup 1 > $out/write.log 2>&1
cd $repo
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 scripts/pmc_summary.py $out/fetch $out/write $out/pmc_hbm.csv
rm -rf $out/stats $out/fetch $out/write

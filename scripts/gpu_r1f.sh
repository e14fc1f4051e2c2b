#!/bin/bash
mkdir -p gpurun_out/r1f
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r1f/prof -o p -- python3 /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/r1f/prof.log 2>&1
cd /root/repo
find gpurun_out/r1f -name "*_kernel_trace.csv" -delete; find gpurun_out/r1f -name "*agent_info*" -delete
python bench.py --no-cpu-baseline --mesh 1024 --steps 3 --warmup 1 > gpurun_out/r1f/bench_1024.json 2>gpurun_out/r1f/bench.err
true

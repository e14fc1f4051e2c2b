#!/bin/bash
mkdir -p gpurun_out/r1e
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r1e/tests.log
python bench.py --no-cpu-baseline > gpurun_out/r1e/bench_plain.json 2> gpurun_out/r1e/bench_plain.err
cd /tmp && export TMPDIR=/tmp
for P in 2 8; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r1e/prof_$P -o p -- python3 /root/repo/scripts/mr_probe.py --ranks $P --steps 5 --warmup 1 > /root/repo/gpurun_out/r1e/prof_$P.log 2>&1
done
cd /root/repo
find gpurun_out/r1e -name "*_kernel_trace.csv" -delete; find gpurun_out/r1e -name "*agent_info*" -delete
true

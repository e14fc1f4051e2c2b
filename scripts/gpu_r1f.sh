#!/bin/bash
mkdir -p gpurun_out/r1f
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r1f/tests.log
python bench.py --no-cpu-baseline > gpurun_out/r1f/bench.json 2> gpurun_out/r1f/bench.err
python bench.py --no-cpu-baseline --mesh 1024 --steps 3 --warmup 1 > gpurun_out/r1f/bench_1024.json 2> gpurun_out/r1f/bench_1024.err
python bench.py --no-cpu-baseline --window tsc --dtype f4 --gradient 0 > gpurun_out/r1f/bench_c3.json 2> gpurun_out/r1f/bench_c3.err
python bench.py --no-cpu-baseline --mesh 256 > gpurun_out/r1f/bench_256.json 2> gpurun_out/r1f/bench_256.err
true

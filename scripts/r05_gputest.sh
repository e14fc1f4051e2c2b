# the whole GPU suite + smoke + the default bench line of the current build (logs under gpurun_out/$1)
out=gpurun_out/${1:-r05_t}; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu > $out/gputest.log 2>&1; tail -5 $out/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 1500 $out/bench_default.json

out=gpurun_out/r05_f; mkdir -p $out
timeout 900 python -m pytest tests/test_binned.py tests/test_pm.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
export PMESH_AMD_BENCH_NOCHECK=1 PMESH_AMD_BENCH_STEPS=1
for lib in lb1 lb2 lb3; do
  export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so
  for cfg in "" "--config c3"; do
    echo "[$lib] $cfg"; python bench.py $cfg --no-cpu-baseline --steps 3 --warmup 2 2>&1 | grep '^step' | tail -2
  done
done 2>&1 | tee $out/leanbin_split.txt

for lib in "" cf2 cf1; do
  if [ -n "$lib" ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so; else unset PMESH_AMD_LIBRARY; fi
  for cfg in "--data clustered" "--data clustered --window pcs --mass array" "--data clustered --window tsc --dtype f4"; do
    python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-8s] %-50s %.3f' % ('$lib', '$cfg', d['ms_per_step']), {k: round(v,3) for k,v in d['stages_ms'].items()})"
  done
done

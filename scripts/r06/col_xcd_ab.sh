#!/bin/bash
# A/B of the XCD-aware tile order of the column passes (the product: PMX_COL_XCD=1; without: scripts/build_variant.sh xcd0 "-DPMX_COL_XCD=0"
# pmx_colfft.hip) against tile = blockIdx: the column micro (padded and dense lines), the headline, 8 thread ranks.
out=gpurun_out/${1:-r06_colxcd}; mkdir -p $out
for lib in xcd0 product xcd0 product; do
  if [ $lib = xcd0 ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_xcd0.so; else unset PMESH_AMD_LIBRARY; fi
  echo "== $lib"
  timeout 300 python scripts/r06/col_micro.py 2 2>&1 | grep -v amdgpu.ids
  for cfg in "" "--mesh 1024 --steps 5 --warmup 2" "--config c3"; do
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && python - $out/r.json "[$lib] $cfg" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-50s %8.3f ms  r2c %.3f c2r %.3f" % (sys.argv[2], d["ms_per_step"], st["r2c"], st["c2r"]))
PY
  done
done
for lib in xcd0 product; do
  if [ $lib = xcd0 ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_xcd0.so; else unset PMESH_AMD_LIBRARY; fi
  echo "== $lib: 8 thread ranks"
  scripts/mr_kstats.sh ${1:-r06_colxcd}_slab_$lib --ranks 8 --mesh 512 --steps 12 --warmup 2 2>&1 | grep "pmx kernels\|fft"
  scripts/mr_kstats.sh ${1:-r06_colxcd}_pencil_$lib --ranks 8 --mesh 512 --steps 12 --warmup 2 --np 2x4 --migrate 1 2>&1 | grep "pmx kernels\|fft"
done

#!/bin/bash
# build-time knobs of the column / round-trip FFT kernels against the product (variant libraries, same box)
out=gpurun_out/${1:-r06_ffttune}; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-44s %8.3f ms  r2c %.3f c2r %.3f" % (sys.argv[2], d["ms_per_step"], st["r2c"], st["c2r"]))
PY
}
for rep in 1 2; do
for cfg in "--config c3" "--dtype f4" "" "--mesh 256"; do
  for lib in product $VARIANTS; do
    if [ $lib = product ]; then unset PMESH_AMD_LIBRARY; else export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so; fi
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && line $out/r.json "[$lib] $cfg"
  done
done
done

#!/bin/bash
# colfft_round_kernel with its first / last passes in registers for float and for N = 1024 (the forms the compiler spills:
# PMX_ROUND_REGS_F4=1, PMX_ROUND_REGS_MAXLOG=10; scripts/build_variant.sh regsf4 "..." pmx_colfft.hip) against the product
out=gpurun_out/${1:-r06_roundregs}; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-50s %8.3f ms  r2c %.3f c2r %.3f" % (sys.argv[2], d["ms_per_step"], st["r2c"], st["c2r"]))
PY
}
for rep in 1 2; do
for cfg in "--config c3" "--dtype f4" "--mesh 1024 --steps 5 --warmup 2" "--mesh 1024 --dtype f4 --steps 5 --warmup 2" ""; do
  for lib in product regsf4; do
    if [ $lib = regsf4 ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_regsf4.so; else unset PMESH_AMD_LIBRARY; fi
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && line $out/r.json "[$lib] $cfg"
  done
done
done

#!/bin/bash
# threads per workgroup of the readout on float canvases (PMX_TILE_THREADS_RF4: 256 in the product) — variant libraries, same box
out=gpurun_out/${1:-r06_rf4}; mkdir -p $out
line() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st=d["stages_ms"]
print("%-44s %8.3f ms  readout %.3f" % (sys.argv[2], d["ms_per_step"], st["readout"]))
PY
}
for rep in 1 2; do
for cfg in "--config c3" "--dtype f4" "--dtype f4 --window pcs"; do
  for lib in product rf4_512 rf4_128; do
    if [ $lib = product ]; then unset PMESH_AMD_LIBRARY; else export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so; fi
    timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > $out/r.json 2> $out/r.err && line $out/r.json "[$lib] $cfg"
  done
done
done

out=gpurun_out/r05_d; mkdir -p $out
run() {
    timeout 300 python bench.py $2 --no-cpu-baseline --steps 10 --warmup 3 > $out/r.json 2>$out/r.err || tail -3 $out/r.err
    python - "$1" "$2" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/r05_d/r.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[%-8s] %-45s %.3f ms  bin %.2f paint %.3f r2c %.2f c2r %.2f readout %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
except Exception as e: print(sys.argv[1:], 'failed', e)
PY
}
for lib in "" $VARIANTS; do
  if [ -n "$lib" ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so; [ -f $PMESH_AMD_LIBRARY ] || continue; else unset PMESH_AMD_LIBRARY; fi
  for cfg in "--config c3" "--window pcs --dtype f4" "--window tsc --dtype f4 --data clustered"; do run "${lib:-product}" "$cfg"; done
done 2>&1 | tee $out/stages.txt

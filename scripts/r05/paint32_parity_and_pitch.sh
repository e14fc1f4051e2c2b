# r05: 32-bit regions — parity first, then the stage times of the float-canvas configurations, product and variants
out=gpurun_out/r05_b; mkdir -p $out
timeout 900 python -m pytest tests/test_binned.py tests/test_window.py tests/test_halo_defer.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
for lib in "" p34t256 p36 p48; do
  for cfg in "--config c3" "--window pcs --dtype f4" "--window tsc --dtype f4 --data clustered" "--window pcs --dtype f4 --data clustered"; do
    if [ -n "$lib" ]; then export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_$lib.so; [ -f $PMESH_AMD_LIBRARY ] || continue; else unset PMESH_AMD_LIBRARY; fi
    timeout 300 python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 > $out/r.json 2>$out/r.err || tail -3 $out/r.err
    python - "$lib" "$cfg" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/r05_b/r.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[%-8s] %-45s %.3f ms  bin %.2f paint %.3f r2c %.2f c2r %.2f readout %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
except Exception as e: print(sys.argv[1:], 'failed', e)
PY
  done
done 2>&1 | tee $out/variants.txt

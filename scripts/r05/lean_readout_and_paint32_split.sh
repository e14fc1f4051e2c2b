# r05: lean readout — parity, then stage times; the paint32 kernel without its deposit loop (experiment build, no bench line)
out=gpurun_out/r05_c; mkdir -p $out
timeout 900 python -m pytest tests/test_binned.py tests/test_window.py tests/test_pm.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() {
    timeout 300 python bench.py $2 --no-cpu-baseline --steps 10 --warmup 3 > $out/r.json 2>$out/r.err || tail -3 $out/r.err
    python - "$1" "$2" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/r05_c/r.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[%-8s] %-45s %.3f ms  bin %.2f paint %.3f r2c %.2f c2r %.2f readout %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
except Exception as e: print(sys.argv[1:], 'failed', e)
PY
}
for cfg in "" "--config c3" "--window tsc" "--window pcs" "--dtype f4" "--window pcs --dtype f4" "--data clustered" "--window pcs --data clustered --mass array" "--mesh 256"; do run product "$cfg"; done 2>&1 | tee $out/stages.txt
export PMESH_AMD_LIBRARY=$PWD/pmesh_amd/libpmesh_amd_nodep32.so PMESH_AMD_BENCH_NOCHECK=1
echo 'paint32 without its deposit loop (experiment build):' | tee -a $out/stages.txt
PMESH_AMD_BENCH_STEPS=1 python bench.py --config c3 --no-cpu-baseline --steps 4 --warmup 2 2>&1 | grep '^step' | tee -a $out/stages.txt

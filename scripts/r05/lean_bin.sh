out=gpurun_out/r05_e; mkdir -p $out
timeout 900 python -m pytest tests/test_binned.py tests/test_pm.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() {
    timeout 300 python bench.py $2 --no-cpu-baseline --steps 10 --warmup 3 > $out/r.json 2>$out/r.err || tail -3 $out/r.err
    python - "$1" "$2" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/r05_e/r.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[%-8s] %-45s %.3f ms  bin %.3f paint %.3f r2c %.2f c2r %.2f readout %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
except Exception as e: print(sys.argv[1:], 'failed', e)
PY
}
for cfg in "" "--config c3" "--window tsc" "--window pcs" "--dtype f4" "--data clustered" "--window pcs --data clustered --mass array" "--mesh 256" "--drift 0.5" "--drift 2.0" "--mesh 1024"; do run product "$cfg"; done 2>&1 | tee $out/stages.txt

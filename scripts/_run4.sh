timeout 900 python -m pytest tests/test_halo_defer.py -x -q 2>&1 | tail -5
for cfg in "--mesh 1024 --steps 5" "--mesh 1024 --double 1 --mass array --window pcs --data clustered --steps 3 --warmup 1"; do
  for of in 1 0; do
    timeout 600 python bench.py $cfg --out-field $of --no-cpu-baseline > gpurun_out/v.json 2>gpurun_out/v.err
    python - "$cfg" $of <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/v.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[out_field=%s] %-40s %.3f ms  bin %.2f paint %.3f r2c %.3f c2r %.3f readout %.2f host %.2f" % (sys.argv[2], sys.argv[1][:40], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"], d["host_issue_ms_per_step"]))
except Exception as ex:
    print("FAILED", sys.argv[1:], open("gpurun_out/v.err").read()[-1500:])
PY
  done
done

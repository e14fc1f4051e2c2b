timeout 900 python -m pytest tests/test_halo_defer.py -x -q 2>&1 | tail -5
for cfg in "" "--window tsc --dtype f4 --gradient 0" "--window pcs"; do
  for of in 1 0 2; do
    if [ $of = 2 ]; then export PMESH_AMD_HALO_DEFER=always; off=1; else unset PMESH_AMD_HALO_DEFER; off=$of; fi
    timeout 300 python bench.py $cfg --out-field $off --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/v.json 2>gpurun_out/v.err
    python - "$cfg" $of <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/v.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[mode=%s] %-40s %.3f ms  bin %.2f paint %.3f r2c %.3f c2r %.3f readout %.2f host %.2f" % (sys.argv[2], sys.argv[1], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"], d["host_issue_ms_per_step"]))
except Exception as ex:
    print("FAILED", sys.argv[1:], open("gpurun_out/v.err").read()[-1500:])
PY
  done
done

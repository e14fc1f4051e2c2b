for mb in 0 128 160 192 224; do
  for of in 1 0; do
    PMESH_AMD_L3_BLOCK_MB=$mb timeout 300 python bench.py $cfg --out-field $of --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/v.json 2>gpurun_out/v.err
    python - "L3=$mb" $of <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/v.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
    print("[out_field=%s] %-40s %.3f ms  bin %.2f paint %.3f r2c %.3f c2r %.3f readout %.2f" % (sys.argv[2], sys.argv[1], d["ms_per_step"], st["bin"], st["paint"], st["r2c"], st["c2r"], st["readout"]))
except Exception as ex:
    print("FAILED", sys.argv[1:], open("gpurun_out/v.err").read()[-1500:])
PY
  done
done

for cfg in "--window tsc --dtype f4 --gradient 0" "--window pcs" "" "--window tsc"; do
for srt in auto always; do
PMESH_AMD_SORTED=$srt timeout 300 python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/v.json 2>gpurun_out/v.err
python - "$srt" "$cfg" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/v.json").read().strip().splitlines()[-1]); st=d["stages_ms"]
print("[sorted=%-6s] %-42s %.3f ms  bin %.2f paint %.2f readout %.2f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], st["bin"], st["paint"], st["readout"]))
PY
done; done

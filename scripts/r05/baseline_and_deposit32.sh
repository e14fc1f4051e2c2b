mkdir -p gpurun_out/r05_a
scripts/deposit32_model > gpurun_out/r05_a/deposit32_model.txt 2>&1
for cfg in "" "--config c3" "--window pcs" "--window tsc" "--dtype f4" "--window pcs --dtype f4" "--data clustered" "--mesh 256"; do
  echo "== $cfg" >> gpurun_out/r05_a/bench.txt
  timeout 300 python bench.py $cfg --no-cpu-baseline --steps 10 --warmup 3 2>gpurun_out/r05_a/err.txt | tail -1 >> gpurun_out/r05_a/bench.txt
done
cat gpurun_out/r05_a/deposit32_model.txt
python - <<'PY'
import json
for l in open('gpurun_out/r05_a/bench.txt'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); st=d['stages_ms']; print('  %.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in st.items()})
    except Exception as e: print('  ?', l[:200])
PY

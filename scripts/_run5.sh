bash scripts/kstats.sh ks256 --mesh 256 --steps 20 --warmup 5 2>&1 | head -30
timeout 300 python bench.py --mesh 256 --no-cpu-baseline --steps 50 --warmup 10 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['host_issue_ms_per_step'])"

#!/bin/bash
# A/B of compile-time switches of pmx_colfft.hip on the GPU box:
#   scripts/ab_fft.sh "<bench args>" "<flags A>" "<flags B>" ...
args=$1; shift
for flags in "$@"; do
  (cd pmesh_amd/csrc && touch pmx_colfft.hip && make EXTRA="$flags" 2>&1 | grep -E " error" | head -3)
  timeout 300 python bench.py $args --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flags] $args', round(d['ms_per_step'],3), 'r2c', d['stages_ms']['r2c'], 'c2r', d['stages_ms']['c2r'])"
done

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j.get('roofline') or {}; print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s frac', round(r.get('frac') or 0,3), {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items() if k in ('fused_fwd_bwd','item_update','user_update')}, j.get('plan_ms'))"; }
for m in none first:32 stride8:32 first:64 stride8:64 first:16 stride8:16 first:128; do
  if [ $m = none ]; then timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | line "mask=$m";
  else RC_SIDE_CUMASK=$m timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | line "mask=$m"; fi
done
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | line "mask=none(again)"

#!/bin/bash
TAG=${1:-bp}; ALT=${2:-RC_STEP_UB=0}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_bprmf.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -4
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j.get('roofline') or {}; print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s frac', round(r.get('frac') or 0,3), (r.get('alone') or {}).get('frac'), {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, j.get('phases_alone_ms'))"; }
for i in 1 2 3; do
timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>$OUT/bench.err | tee $OUT/bench.json | line A
env $ALT timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_alt.json | line "B($ALT)"
done
timeout 300 python bench.py --steps 30 --warmup 5 --opt Adam --no-cpu-baseline 2>/dev/null | line A_adam
env $ALT timeout 300 python bench.py --steps 30 --warmup 5 --opt Adam --no-cpu-baseline 2>/dev/null | line "B_adam($ALT)"

#!/bin/bash
# usage (on the GPU box): bash tools/ab_step.sh  -> bench lines for env-variable variants of the step (no rebuild)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$tag', round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['phases_ms'].items()})"
}
run plan_default RC_X=0
run plan_shift12 RC_PLAN_SHIFT=12
run plan_shift11 RC_PLAN_SHIFT=11
run sort_pipeline RC_BPRMF_STEP=sort

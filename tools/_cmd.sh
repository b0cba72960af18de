run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$tag', round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['phases_ms'].items()}, round(j['roofline'].get('alone',{}).get('avg_ms',0),4))"
}
run default RC_X=0
run shift12 RC_PLAN_SHIFT=12
run shift11 RC_PLAN_SHIFT=11
run shift10 RC_PLAN_SHIFT=10
run serial RC_BPRMF_STEP=serial

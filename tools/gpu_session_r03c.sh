#!/bin/bash
# Round 3, session c: how well the look-ahead plan hides beside the row updates -- stream priority, what the look-ahead
# prepares, occupancy cap of the row-update kernel (A/B inside one call: boxes differ by +-7 %).
TAG=${1:-r03c}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$tag', round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['phases_ms'].items() if k in ('fused_fwd_bwd','item_update','user_update','total')}, j.get('plan_ms'))"
}
{
run base RC_X=0
run prio RC_SIDE_PRIO=high
run front RC_AHEAD_PART=front
run front_prio RC_AHEAD_PART=front RC_SIDE_PRIO=high
run early_prio RC_AHEAD_FORK=early RC_SIDE_PRIO=high
for t in rows_w6 rows_w5 rows_w4; do
  run $t RC_LIB_PATH=$PWD/tools/bin/lib_$t.so
  run ${t}_prio RC_LIB_PATH=$PWD/tools/bin/lib_$t.so RC_SIDE_PRIO=high
done
run base_again RC_X=0
} 2>&1 | tee $OUT/ab_overlap.txt

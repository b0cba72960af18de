#!/bin/bash
# usage (on the GPU box): bash tools/exp_variants.sh "RC_A" "RC_B RC_C" ...   -> one bench line per -D set
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/variants
run() {
  tag=$1; shift
  python -m rechorus_amd.csrc.build "$@" > gpurun_out/variants/build_$tag.log 2>&1 || { echo "build failed $tag"; tail -5 gpurun_out/variants/build_$tag.log; return; }
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$tag', round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['phases_ms'].items()})"
}
run base
for defs in "$@"; do
  args=""; for d in $defs; do args="$args --define $d"; done
  run "$(echo $defs | tr ' ' '+')" $args
done
python -m rechorus_amd.csrc.build > /dev/null 2>&1

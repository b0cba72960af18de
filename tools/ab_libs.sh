#!/bin/bash
# usage (GPU box): bash tools/ab_libs.sh tag1 tag2 ...  -> one line per experimental library tools/bin/lib_<tag>.so (base = in-tree)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$tag', round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['phases_ms'].items()})"
}
run base RC_X=0
for t in "$@"; do run $t RC_LIB_PATH=$PWD/tools/bin/lib_$t.so; done

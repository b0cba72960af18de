#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,3), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()})"; }
for cfg in "256 20" "256 50" "512 20" "512 32" "128 20"; do
set -- $cfg
RC_SASREC_IMPL=sequence timeout 300 python bench.py --workload sasrec --batch $1 --hist $2 --steps 200 --no-cpu-baseline 2>/dev/null | line "b$1 L$2 sequence"
RC_SASREC_IMPL=batch timeout 300 python bench.py --workload sasrec --batch $1 --hist $2 --steps 200 --no-cpu-baseline 2>/dev/null | line "b$1 L$2 batch"
done
timeout 600 python tools/bench_plugin_epoch.py 2>/dev/null | tail -c 1200
RC_SASREC_IMPL=batch timeout 600 python tools/bench_plugin_epoch.py 2>/dev/null | tail -c 1200

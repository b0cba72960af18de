#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-adam}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sasrec.py tests/test_gpu_plugin.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -15
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,3), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()})"; }
timeout 300 python bench.py --workload sasrec --opt Adam --no-cpu-baseline 2>$OUT/adam.err | tee $OUT/bench_sasrec_adam.json | line "adam graph"
RC_SAS_GRAPH=0 timeout 300 python bench.py --workload sasrec --opt Adam --no-cpu-baseline 2>/dev/null | line "adam eager"
timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>/dev/null | line "sgd graph"
tail -3 $OUT/adam.err
timeout 600 python tools/bench_plugin_epoch.py > $OUT/plugin_epoch.json 2> $OUT/plugin_epoch.err
grep SASRec $OUT/plugin_epoch.json | cut -c1-260
tail -3 $OUT/plugin_epoch.err

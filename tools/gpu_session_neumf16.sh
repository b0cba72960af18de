#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_neumf.py tests/test_gpu_plan.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -6
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,3), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, j.get('phases_tflops'), j.get('head_fwd_gather_gbps'))"; }
timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>/dev/null | line "A bwd16"
RC_NEUMF_BWD16=0 timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>/dev/null | line "B bwd one kernel"

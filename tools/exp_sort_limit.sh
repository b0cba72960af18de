#!/bin/bash
# merge-sort limit of the id sort vs step time at several batch sizes (run on the GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lim in 1048576 131072 32768 8192; do
  python -m rechorus_amd.csrc.build --define RC_MERGE_SORT_LIMIT=$lim > /dev/null 2>&1 || { echo build failed; continue; }
  for b in 256 1024 2048 8192; do
    timeout 100 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('limit', $lim, 'B', $b, 'ms/step', round(j['ms_per_step'],4), 'Mtuples/s', round(j['value']/1e6,2))"
  done
done
python -m rechorus_amd.csrc.build > /dev/null 2>&1

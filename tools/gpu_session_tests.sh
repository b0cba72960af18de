#!/bin/bash
# full GPU suite + plugin epoch: $1 = tag
TAG=${1:-t}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout 600 python tools/bench_plugin_epoch.py > $OUT/plugin_epoch.json 2> $OUT/plugin_epoch.err
python - <<PY
import json
for r in json.load(open("$OUT/plugin_epoch.json")) if open("$OUT/plugin_epoch.json").read().strip().startswith("[") else []:
    print(r.get("config"), r.get("epoch_s"), r.get("tuples_per_s"), r.get("dev"))
PY
tail -c 600 $OUT/plugin_epoch.json

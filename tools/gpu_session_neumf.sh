#!/bin/bash
TAG=${1:-nmf}; ALT=${2:-RC_NEUMF_OVERLAP=0}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_neumf.py tests/test_gpu_plan.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider -k "neumf or Neumf" 2>&1 | tail -4
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j.get('roofline') or {}; print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, j.get('phases_tflops'))"; }
for i in 1 2; do
timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>$OUT/neumf.err | tee $OUT/bench_neumf.json | line A
env $ALT timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>/dev/null | tee $OUT/bench_neumf_alt.json | line "B($ALT)"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_neumf -o kt --output-format csv -- \
  python $R/bench.py --workload neumf --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_neumf.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
python - <<PY
import csv,glob
for f in glob.glob("$OUT/prof_neumf/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY

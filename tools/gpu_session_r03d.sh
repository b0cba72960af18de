#!/bin/bash
TAG=${1:-r03d}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_sasrec.py tests/test_gpu_bprmf.py tests/test_gpu_plugin.py tests/test_gpu_pipeline.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()})"; }
timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>$OUT/sasrec.err | tee $OUT/bench_sasrec.json | line sasrec_plan
RC_TABLE_UPDATE=sort timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>/dev/null | tee $OUT/bench_sasrec_sort.json | line sasrec_sort
run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$tag', round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['phases_ms'].items() if k in ('fused_fwd_bwd','item_update','user_update','total')})"
}
{
run base RC_X=0
run front RC_AHEAD_PART=front
run early RC_AHEAD_FORK=early
run early_front RC_AHEAD_PART=front RC_AHEAD_FORK=early
run base2 RC_X=0
run front2 RC_AHEAD_PART=front
run early2 RC_AHEAD_FORK=early
run early_front2 RC_AHEAD_PART=front RC_AHEAD_FORK=early
} 2>&1 | tee $OUT/ab_overlap2.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_sasrec -o kt --output-format csv -- \
  python $R/bench.py --workload sasrec --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_sasrec.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
python - <<PY
import csv,glob
for f in glob.glob("$OUT/prof_sasrec/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:40]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY

#!/bin/bash
# Round 3, session b: hashed plan + plan consumers, BPRMF step after the XCD-contiguous tiles / folded chunk pass,
# NeuMF / SASRec lines with the plan-driven table updates (A/B against the radix sort).
TAG=${1:-r03b}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_neumf.py tests/test_gpu_sasrec.py tests/test_gpu_sharded.py tests/test_gpu_bprmf.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, j.get('plan_ms'), (j.get('roofline') or {}).get('frac'))"; }
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$OUT/bench.err | tee $OUT/bench.json | line bprmf
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --opt Adam 2>/dev/null | tee $OUT/bench_adam.json | line bprmf_adam
timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>$OUT/neumf.err | tee $OUT/bench_neumf.json | line neumf_plan
RC_TABLE_UPDATE=sort timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>/dev/null | tee $OUT/bench_neumf_sort.json | line neumf_sort
timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>$OUT/sasrec.err | tee $OUT/bench_sasrec.json | line sasrec_plan
RC_TABLE_UPDATE=sort timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>/dev/null | tee $OUT/bench_sasrec_sort.json | line sasrec_sort
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o kt --output-format csv -- \
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_neumf -o kt --output-format csv -- \
  python $R/bench.py --workload neumf --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_neumf.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
python - <<PY
import csv,glob
for d in ("prof","prof_neumf"):
    for f in glob.glob("$OUT/"+d+"/**/*kernel_stats.csv", recursive=True):
        print("==",d)
        for r in list(csv.DictReader(open(f)))[:14]:
            print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
bash tools/pmc_collect.sh $TAG/pmc 2>&1 | tail -16

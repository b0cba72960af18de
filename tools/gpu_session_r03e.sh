#!/bin/bash
TAG=${1:-r03e}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_neumf.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j.get('roofline') or {}; print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, 'alone', j.get('phases_alone_ms'), j.get('plan_ms'), 'frac', r.get('frac'), (r.get('alone') or {}).get('frac'), 'eff', j.get('step_effective_gbps'))"; }
timeout 600 python bench.py --steps 40 --warmup 8 2>$OUT/bench.err | tee $OUT/bench.json | line bprmf
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --opt Adam 2>/dev/null | tee $OUT/bench_adam.json | line bprmf_adam
timeout 300 python bench.py --steps 100 --warmup 10 --batch 8192 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_b8192.json | line bprmf_b8192
timeout 300 python bench.py --steps 300 --warmup 20 --batch 256 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_b256.json | line bprmf_b256
timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>$OUT/neumf.err | tee $OUT/bench_neumf.json | line neumf_plan
RC_TABLE_UPDATE=sort timeout 300 python bench.py --workload neumf --no-cpu-baseline 2>/dev/null | tee $OUT/bench_neumf_sort.json | line neumf_sort
timeout 600 python bench.py --workload neumf --items 100000001 --users 10000001 --steps 30 --warmup 5 2>$OUT/neumf100m.err | tee $OUT/bench_neumf_100M.json | line neumf_100M
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
        for r in list(csv.DictReader(open(f)))[:16]:
            print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
bash tools/pmc_collect.sh $TAG/pmc 2>&1 | tail -16

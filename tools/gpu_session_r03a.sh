#!/bin/bash
# Round 3, first GPU session: parity of the restructured BPRMF step, bench line, A/B of store / load flavours and of the
# look-ahead fork point, kernel trace, PMC passes.   usage: gpurun --timeout 1500 -- 'bash tools/gpu_session_r03a.sh'
TAG=${1:-r03a}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
{ rocm-smi --showproductname 2>/dev/null | head -20; nproc; free -g | head -2; lscpu | grep "Model name"; } > $OUT/env.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
echo "--- A/B"
export BENCH_ARGS=""
bash tools/ab_libs.sh pu_plainld pu_plainst pu_sc1st fu_sc1st both_sc1st 2>&1 | tee $OUT/ab.txt
echo "early-fork $(RC_AHEAD_FORK=early timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],4), {k:round(v,4) for k,v in j['phases_ms'].items()})")" | tee -a $OUT/ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o kt --output-format csv -- \
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
python - <<PY
import csv,glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
bash tools/pmc_collect.sh $TAG/pmc 2>&1 | tail -30

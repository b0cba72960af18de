#!/bin/bash
# quick SASRec A/B: $1 = tag, $2 = env assignment for the B leg (e.g. RC_SAS_ROWS16=0), $3 = optional pytest -k filter
TAG=${1:-q}; ALT=${2:-RC_SAS_ROWS16=0}; KF=${3:-}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -n "$KF" ]; then
timeout 900 python -m pytest tests/test_gpu_sasrec.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider -k "$KF" 2>&1 | tail -5
else
timeout 900 python -m pytest tests/test_gpu_sasrec.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -5
fi
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, j.get('phases_tflops'))"; }
for i in 1 2; do
timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>$OUT/sasrec.err | tee $OUT/bench_sasrec.json | line A
env $ALT timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>/dev/null | tee $OUT/bench_sasrec_alt.json | line "B($ALT)"
done
RC_SAS_OVERLAP=0 timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>/dev/null | tee $OUT/bench_sasrec_onestream.json | line A_onestream
timeout 300 python bench.py --workload sasrec --batch 256 --steps 200 --no-cpu-baseline 2>/dev/null | line A_b256
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_sasrec -o kt --output-format csv -- \
  python $R/bench.py --workload sasrec --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_sasrec.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
python - <<PY
import csv,glob
for f in glob.glob("$OUT/prof_sasrec/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY

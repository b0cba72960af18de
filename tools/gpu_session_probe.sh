#!/bin/bash
# $1 = tag; SASRec tests, host-launch probe (eager / graph), bench A/B, kernel trace
TAG=${1:-p}; ALT=${2:-RC_SAS_LAST_ROW=1}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sasrec.py tests/test_gpu_plan.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -5
timeout 300 python tools/sasrec_host_probe.py 2>&1 | grep -v amdgpu.ids | tail -7
B=256 timeout 300 python tools/sasrec_host_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, j.get('phases_tflops'))"; }
timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>$OUT/sasrec.err | tee $OUT/bench_sasrec.json | line A
env $ALT timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>/dev/null | tee $OUT/bench_sasrec_alt.json | line "B($ALT)"
RC_SAS_OVERLAP=0 timeout 300 python bench.py --workload sasrec --no-cpu-baseline 2>/dev/null | tee $OUT/bench_sasrec_onestream.json | line A_onestream
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_sasrec -o kt --output-format csv -- \
  python $R/bench.py --workload sasrec --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_sasrec.log 2>&1
cd $R
python tools/trace_step.py $OUT/prof_sasrec/kt_kernel_trace.csv sb_lr_headT_kernel\<64,\ 2 -3 | cut -c1-120 | head -34

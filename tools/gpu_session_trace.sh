#!/bin/bash
# kernel trace of one bench line: $1 = tag, rest = bench.py arguments
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 300 python bench.py "$@" --no-cpu-baseline 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o kt --output-format csv -- \
  python $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +30M -delete 2>/dev/null
ls -la $OUT/prof

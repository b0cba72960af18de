#!/bin/bash
# kernel times of library variants: $1 = tag, $2.. = variant tags (tools/bin/lib_<v>.so); "std" = the standard library
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  LIB=""; [ "$v" != std ] && LIB=$R/tools/bin/lib_$v.so
  cd /tmp
  RC_LIB_PATH=$LIB timeout 200 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$v -o kt --output-format csv -- \
    python $R/bench.py --workload sasrec --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_$v.log 2>&1
  cd $R
  rm -f $OUT/prof_$v/*kernel_trace.csv
  python - <<PY
import csv,glob
for f in glob.glob("$OUT/prof_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sb_lr_" in r["Name"] or "seg_rows" in r["Name"]:
            print("$v", r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done

#!/bin/bash
# usage (GPU box): bash tools/prof_step.sh <outdir> [bench args]  -> rocprofv3 kernel stats of bench.py (top kernels printed)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > $R/$OUT/prof.log 2>&1
cd $R
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
python - <<PY
import csv,glob
f=glob.glob("$OUT/prof/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n=r["Name"]
    if "at::native" in n: continue
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:90]}')
PY

#!/bin/bash
# usage (GPU box): bash tools/prof_cmd.sh <outdir> <python script + args...>  -> rocprofv3 kernel stats (rc:: kernels printed)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o kt --output-format csv -- python "$@" > $R/$OUT/prof.log 2>&1
cd $R
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
python - <<PY
import csv,glob
fs=glob.glob("$OUT/prof/*kernel_stats.csv")
if not fs: print("no stats (see $OUT/prof.log)")
else:
    for r in list(csv.DictReader(open(fs[0]))):
        n=r["Name"]
        if "rc::" in n: print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:90]}')
PY

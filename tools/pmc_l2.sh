#!/bin/bash
# L2 hit rate per kernel: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum (own pass, kernel-trace only) -> gpurun_out/<tag>/l2.txt
TAG=${1:-pmc}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/l2 -o l --output-format csv -- python $R/tools/pmc_workload.py > $OUT/l2.log 2>&1
cd $R
python - <<PY > $OUT/l2.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/l2/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        acc[row["Kernel_Name"].split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    if "rc::" not in k: continue
    h = max(v.get("TCC_HIT_sum", [0])); m = max(v.get("TCC_MISS_sum", [0]))
    if h + m: print(f"{k:72s} hit {h:12.0f} miss {m:12.0f} rate {h/(h+m):.3f}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/l2.txt; tail -2 $OUT/l2.log

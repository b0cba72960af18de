#!/bin/bash
# rocprofv3 PMC passes (one counter per pass, kernel-trace only) -> gpurun_out/<tag>/pmc.json
TAG=${1:-pmc}
WL=${2:-}          # extra arguments of the workload, e.g. "--workload neumf"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- python $R/tools/pmc_workload.py $WL > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- python $R/tools/pmc_workload.py $WL > $OUT/write.log 2>&1
cd $R
STEPS=0
case "$WL" in *sasrec*|*deepfm*) STEPS=${PMC_STEPS:-5};; esac      # whole-step mode (tools/pmc_workload.py)
python tools/pmc_summarize.py $OUT/fetch $OUT/write 2560000256 $OUT/pmc.json $STEPS > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/pmc_summary.txt; tail -2 $OUT/fetch.log

# rocprofv3 kernel table of one configuration of tools/bench_sasrec_layers.py:  bash tools/prof_layers.sh <tag> [d blocks heads history B K dropout]
TAG=${1:-r09j}; shift
CFG=${@:-128 2 4 50 4096 99 0.2}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o kt --output-format csv -- python $R/tools/bench_sasrec_layers.py $CFG > $R/gpurun_out/$TAG/run.log 2>&1
find $R/gpurun_out/$TAG/prof -name "*kernel_trace.csv" -size +20M -delete
f=$(find $R/gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -16
tail -1 $R/gpurun_out/$TAG/run.log

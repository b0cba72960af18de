#!/bin/bash
# copy the judged artefacts of a GPU session (tools/gpu_run.sh <tag> ...) from gpurun_out/<tag>/ into profiles/<name>_*:
#   bash tools/collect_profiles.sh <tag> [<name>]      bench lines (*.json), rocprofv3 kernel-stats tables (prof_<x>/), PMC summary,
#                                                      environment, the tail of the GPU test log and the smoke line
TAG=$1; NAME=${2:-$1}
S=gpurun_out/$TAG; P=profiles
for f in $S/*.json; do [ -s "$f" ] && cp $f $P/${NAME}_$(basename $f); done
for d in $S/prof_*/; do [ -d "$d" ] || continue; w=$(basename $d); w=${w#prof_}
  f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${NAME}_${w}_kernel_stats.csv; done
[ -f $S/env.txt ] && cp $S/env.txt $P/${NAME}_env.txt
[ -f $S/pmc/pmc.json ] && { cp $S/pmc/pmc.json $P/${NAME}_pmc.json; cp $S/pmc/pmc_summary.txt $P/${NAME}_pmc_summary.txt; cp $S/pmc/pmc.json $P/pmc_latest.json; }
[ -f $S/pytest_gpu.log ] && { tail -4 $S/pytest_gpu.log; [ -f $S/smoke.log ] && tail -1 $S/smoke.log; } > $P/${NAME}_pytest_gpu_tail.txt
rm -f $P/${NAME}_.last_call.json
ls $P | grep "^${NAME}_" | wc -l

#!/bin/bash
# copy the judged artefacts of a full GPU session (tools/gpu_session_r04.sh) from gpurun_out/<tag>/ into profiles/<tag>_*
TAG=$1
S=gpurun_out/$TAG; P=profiles
for f in $S/bench*.json; do cp $f $P/${TAG}_$(basename $f); done
cp $S/prof/kt_kernel_stats.csv $P/${TAG}_kernel_stats.csv
for w in sasrec neumf deepfm; do cp $S/prof_$w/kt_kernel_stats.csv $P/${TAG}_${w}_kernel_stats.csv; done
cp $S/env.txt $P/${TAG}_env.txt
cp $S/plugin_epoch.json $P/${TAG}_plugin_epoch.json
[ -f $S/pmc/pmc.json ] && { cp $S/pmc/pmc.json $P/${TAG}_pmc.json; cp $S/pmc/pmc_summary.txt $P/${TAG}_pmc_summary.txt; }
{ tail -4 $S/pytest_gpu.log; tail -1 $S/smoke.log; } > $P/${TAG}_pytest_gpu_tail.txt
[ -f $S/pmc/pmc.json ] && cp $S/pmc/pmc.json $P/pmc_latest.json
ls $P | grep "^${TAG}_" | wc -l

#!/bin/bash
# SQ counters of the two MFMA kernels (NeuMF head, SASRec encoder): matrix-core busy cycles, LDS bank conflicts,
# wait / issue breakdown.  One rocprofv3 --pmc pass per workload (kernel-trace only), summary to gpurun_out/<tag>/.
TAG=${1:-pmc_mfma}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
if [ "${2:-all}" != "sasrec" ]; then
timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/neumf -o n --output-format csv -- python $R/tools/microbench_neumf.py > $OUT/neumf.log 2>&1
fi
timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/sasrec -o s --output-format csv -- python $R/tools/microbench_sasrec.py > $OUT/sasrec.log 2>&1
cd $R
python - <<PY > $OUT/summary.txt 2>&1
import csv, glob, collections
for wl in ("neumf", "sasrec"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % wl, recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            k = row["Kernel_Name"]
            if "neumf_kernel" in k or "sasrec_fwd_kernel" in k or "sasrec_bwd_kernel" in k or "rc::sb_" in k:
                acc[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        line = {n: round(v) for n, v in m.items()}
        if m.get("SQ_BUSY_CYCLES"):
            line["mfma_busy_over_busy"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / m["SQ_BUSY_CYCLES"], 4)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            line["lds_conflict_frac"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], 4)
        if m.get("SQ_WAVE_CYCLES"):
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                line[n + "_frac"] = round(m.get(n, 0) / m["SQ_WAVE_CYCLES"], 3)
        print(wl, k, line)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
cat $OUT/summary.txt; tail -2 $OUT/sasrec.log

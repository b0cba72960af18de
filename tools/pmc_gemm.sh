#!/bin/bash
# SQ counters of the dense-layer GEMM kernels (tools/gemm_probe.py): matrix-core busy cycles, LDS bank conflicts, wait / issue
# breakdown.  One rocprofv3 --pmc pass (kernel-trace only), summary to gpurun_out/<tag>/summary.txt.
TAG=${1:-pmc_gemm}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/gemm -o g --output-format csv -- python $R/tools/gemm_probe.py > $OUT/gemm.log 2>&1
cd $R
python - <<PY > $OUT/summary.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/gemm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = row["Kernel_Name"]
        if "mlp_gemm" in k:
            acc[k.split("(")[0][-50:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
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
    print(k, line)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
cat $OUT/summary.txt; tail -3 $OUT/gemm.log

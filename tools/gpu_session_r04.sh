#!/bin/bash
# One gpurun call, round 3 (final sessions): GPU parity tests, smoke, bench lines of every workload, rocprofv3 kernel traces, PMC passes.
# Everything lands in gpurun_out/<tag>/.   usage: gpurun --timeout 1800 -- 'bash tools/gpu_session_r03.sh [tag]'
TAG=${1:-r04z}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
{ rocm-smi --showproductname 2>/dev/null | head -20; nproc; free -g | head -2; lscpu | grep "Model name"; } > $OUT/env.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 600 python bench.py --steps 40 --warmup 8 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --steps 30 --warmup 5 --opt Adam --no-cpu-baseline > $OUT/bench_adam.json 2> $OUT/bench_adam.err
timeout 300 python bench.py --steps 100 --warmup 10 --batch 8192 --no-cpu-baseline > $OUT/bench_b8192.json 2> $OUT/bench_b8192.err
timeout 300 python bench.py --steps 300 --warmup 20 --batch 256 --no-cpu-baseline > $OUT/bench_b256.json 2> $OUT/bench_b256.err
timeout 400 python bench.py --workload neumf > $OUT/bench_neumf.json 2> $OUT/bench_neumf.err
timeout 600 python bench.py --workload neumf --items 100000001 --users 10000001 --steps 30 --warmup 5 > $OUT/bench_neumf_100M.json 2> $OUT/bench_neumf_100M.err
RC_NEUMF_FWD16=0 timeout 300 python bench.py --workload neumf --no-cpu-baseline > $OUT/bench_neumf_fwd64.json 2> $OUT/bench_neumf_fwd64.err
timeout 400 python bench.py --workload sasrec > $OUT/bench_sasrec.json 2> $OUT/bench_sasrec.err
timeout 300 python bench.py --workload sasrec --batch 256 --steps 200 --no-cpu-baseline > $OUT/bench_sasrec_b256.json 2> $OUT/bench_sasrec_b256.err
timeout 300 python bench.py --workload sasrec --opt Adam --no-cpu-baseline > $OUT/bench_sasrec_adam.json 2> $OUT/bench_sasrec_adam.err
RC_SAS_GRAPH=0 timeout 300 python bench.py --workload sasrec --no-cpu-baseline > $OUT/bench_sasrec_eager.json 2> $OUT/bench_sasrec_eager.err
RC_SAS_LAST_ROW=0 timeout 300 python bench.py --workload sasrec --no-cpu-baseline > $OUT/bench_sasrec_allrows.json 2> $OUT/bench_sasrec_allrows.err
timeout 300 python bench.py --workload sasrec --layers 2 --no-cpu-baseline > $OUT/bench_sasrec_2layers.json 2> $OUT/bench_sasrec_2layers.err
timeout 300 python bench.py --workload sasrec --hist 100 --no-cpu-baseline > $OUT/bench_sasrec_L100.json 2> $OUT/bench_sasrec_L100.err
timeout 400 python bench.py --workload deepfm > $OUT/bench_deepfm.json 2> $OUT/bench_deepfm.err
timeout 300 python bench.py --workload deepfm --batch 16384 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_deepfm_b16384.json 2> $OUT/bench_deepfm_b16384.err
timeout 300 python bench.py --workload deepfm --batch 131072 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_deepfm_b131072.json 2> $OUT/bench_deepfm_b131072.err
timeout 600 python tools/bench_plugin_epoch.py > $OUT/plugin_epoch.json 2> $OUT/plugin_epoch.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o kt --output-format csv -- \
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_sasrec -o kt --output-format csv -- \
  python $R/bench.py --workload sasrec --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_sasrec.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_neumf -o kt --output-format csv -- \
  python $R/bench.py --workload neumf --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_neumf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_deepfm -o kt --output-format csv -- \
  python $R/bench.py --workload deepfm --batch 131072 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_deepfm.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
[ -n "$SKIP_PMC" ] || bash tools/pmc_collect.sh $TAG/pmc > /dev/null 2>&1
ls -laR $OUT > $OUT/ls.txt 2>&1
tail -5 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
for f in bench bench_adam bench_b8192 bench_b256 bench_neumf bench_neumf_100M bench_neumf_fwd64 bench_sasrec_L100 bench_sasrec bench_sasrec_b256 bench_sasrec_adam bench_sasrec_eager bench_sasrec_allrows bench_sasrec_2layers bench_deepfm bench_deepfm_b16384 bench_deepfm_b131072; do
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$f.json").readline())
    r = j.get("roofline") or {}
    print("$f", round(j["ms_per_step"], 4), "ms", round(j["value"] / 1e6, 3), "M/s roofline", r.get("kernel", "")[:24], round(r.get("frac") or 0, 3), (r.get("alone") or {}).get("frac"), j.get("phases_ms"), j.get("step_effective_gbps"))
except Exception as e:
    print("$f FAILED", e, open("$OUT/$f.err").read()[-400:])
PY
done
cat $OUT/pmc/pmc_summary.txt 2>/dev/null | head -16
tail -c 1500 $OUT/plugin_epoch.json

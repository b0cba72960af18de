#!/bin/bash
TAG=${1:-sm}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_deepfm.py tests/test_gpu_plugin.py tests/test_gpu_bprmf.py tests/test_gpu_neumf.py tests/test_gpu_sasrec.py tests/test_gpu_pipeline.py tests/test_gpu_impression.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -8
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j.get('roofline') or {}; print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,3), 'M/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()})"; }
for i in 1 2; do
timeout 300 python bench.py --workload deepfm --steps 100 --warmup 10 --no-cpu-baseline 2>$OUT/deepfm.err | tee $OUT/bench_deepfm.json | line A_deepfm_b1024
RC_EDB_SMALL=0 timeout 300 python bench.py --workload deepfm --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_deepfm_alt.json | line B_deepfm_b1024_sortroute
done
timeout 600 python tools/bench_plugin_epoch.py > $OUT/plugin_epoch.json 2> $OUT/plugin_epoch.err
RC_EDB_SMALL=0 timeout 600 python tools/bench_plugin_epoch.py > $OUT/plugin_epoch_alt.json 2> $OUT/plugin_epoch_alt.err
python - <<PY
import json
for f in ("plugin_epoch", "plugin_epoch_alt"):
    try:
        j = json.loads(open("$OUT/%s.json" % f).readline())
        for r in j["runs"]:
            print(f, r["config"][:60], r["epoch_s"][1:], r.get("dev"))
    except Exception as e:
        print(f, "FAILED", e)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_deepfm -o kt --output-format csv -- \
  python $R/bench.py --workload deepfm --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_deepfm.log 2>&1
cd $R
python - <<PY
import csv,glob
for f in glob.glob("$OUT/prof_deepfm/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("kernels per step", sum(int(r["Calls"]) for r in rows) / 25.0)
    for r in rows[:12]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY

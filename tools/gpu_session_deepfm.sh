#!/bin/bash
# DeepFM / dense-layer session: $1 = tag, $2 = env assignment of the B leg
TAG=${1:-dfm}; ALT=${2:-RC_MLP_BIG=0}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_deepfm.py tests/test_gpu_plugin.py tests/test_gpu_neumf.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -8
line() { python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j.get('roofline') or {}; print('$1', round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,2), 'M rows/s', {k:round(v,4) for k,v in (j.get('phases_ms') or {}).items()}, 'frac', round(r.get('frac') or 0,3))"; }
for b in 1024 16384 131072; do
  st=50; [ $b = 131072 ] && st=20
  timeout 300 python bench.py --workload deepfm --batch $b --steps $st --warmup 5 --no-cpu-baseline 2>$OUT/deepfm_$b.err | tee $OUT/bench_deepfm_b$b.json | line "A b=$b"
  env $ALT timeout 300 python bench.py --workload deepfm --batch $b --steps $st --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_deepfm_b${b}_alt.json | line "B($ALT) b=$b"
done
cd /tmp
for b in 1024 131072; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_deepfm_$b -o kt --output-format csv -- \
  python $R/bench.py --workload deepfm --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/prof_deepfm_$b.log 2>&1
done
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
python - <<PY
import csv,glob
for f in sorted(glob.glob("$OUT/prof_deepfm_*/**/*kernel_stats.csv", recursive=True)):
    print(f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY

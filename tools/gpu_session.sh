#!/bin/bash
# One gpurun call: GPU parity tests, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tag]'
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
{ rocm-smi --showproductname 2>/dev/null | head -20; nproc; free -g | head -2; lscpu | grep "Model name"; } > $OUT/env.txt 2>&1

timeout 900 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
if ! grep -q "pytest exit 0" $OUT/pytest_gpu.log; then
  # diagnose: same tests with the DPP reductions routed through ds_bpermute
  python -m rechorus_amd.csrc.build --force --no-dpp > $OUT/build_nodpp.log 2>&1
  timeout 600 python -m pytest tests/test_gpu_bprmf.py -m gpu -q --maxfail=10 --tb=line -p no:cacheprovider > $OUT/pytest_gpu_nodpp.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu_nodpp.log
  python -m rechorus_amd.csrc.build --force > $OUT/build_dpp.log 2>&1
fi

timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --steps 100 --warmup 10 --batch 8192 --no-cpu-baseline > $OUT/bench_b8192.json 2> $OUT/bench_b8192.err
timeout 300 python bench.py --steps 300 --warmup 20 --batch 256 --no-cpu-baseline > $OUT/bench_b256.json 2> $OUT/bench_b256.err
timeout 300 python bench.py --steps 30 --warmup 5 --opt Adam --no-cpu-baseline > $OUT/bench_adam.json 2> $OUT/bench_adam.err

cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o kt --output-format csv -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
ls -laR $OUT > $OUT/ls.txt 2>&1
tail -5 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -2; cat $OUT/bench.json; tail -3 $OUT/bench.err

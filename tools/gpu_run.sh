#!/bin/bash
# ONE parameterised GPU session (replaces the per-question gpu_session_*.sh scripts of rounds 1-3).
#   gpurun --timeout 900 -- 'bash tools/gpu_run.sh <tag> <stage> [<stage> ...]'
# Everything lands in gpurun_out/<tag>/.  Stages (any order, run in the order given):
#   tests[:<pytest -k expr or file>]   pytest -m gpu (whole suite, or one file / -k expression)
#   smoke                              __graft_entry__.smoke()
#   bench[:<name>:<bench.py args>]     a bench line -> <name>.json      (plain `bench` = the driver's default command)
#   ab:<name>:<ENV=V,ENV=V>:<args>     the same bench line with environment switches set (A/B inside one call)
#   prof:<name>:<bench.py args>        rocprofv3 --kernel-trace --stats of a bench command -> prof_<name>/
#   profenv:<name>:<ENV=V,..>:<args>   the same with environment switches set
#   pmc                                FETCH_SIZE / WRITE_SIZE passes of the BPRMF step (tools/pmc_collect.sh) -> pmc/
#   py:<name>:<script and args>        python <script> -> <name>.log
TAG=${1:-run}
shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
{ rocm-smi --showproductname 2>/dev/null | head -12; nproc; free -g | head -2; lscpu | grep "Model name"; } > $OUT/env.txt 2>&1

summ() {  # one line per bench json
  python - "$1" <<'PY'
import json, sys
p = sys.argv[1]
try:
    j = json.loads(open(p).readline())
    r = j.get("roofline") or {}
    print(p.split("/")[-1], round(j["ms_per_step"], 4), "ms", round(j["value"] / 1e6, 3), "M/s | roofline", str(r.get("kernel", ""))[:24], r.get("bound"),
          round(r.get("frac") or 0, 3), "| phases", j.get("phases_ms"))
except Exception as e:
    err = p[:-5] + ".err"
    try:
        tail = open(err).read()[-600:]
    except Exception:
        tail = ""
    print(p.split("/")[-1], "FAILED", e, tail)
PY
}

for st in "$@"; do
  kind=${st%%:*}
  rest=${st#*:}
  [ "$rest" == "$st" ] && rest=""
  case $kind in
    tests)
      if [ -z "$rest" ]; then
        timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
      elif [ -f "$rest" ]; then
        timeout 900 python -m pytest "$rest" -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
      else
        timeout 900 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider -k "$rest" > $OUT/pytest_gpu.log 2>&1
      fi
      echo "pytest exit $?" >> $OUT/pytest_gpu.log
      tail -25 $OUT/pytest_gpu.log
      ;;
    tol)       # the whole GPU suite with every comparison recorded -> tol.jsonl -> tolerances.txt (tools/tolerance_report.py)
      rm -f $OUT/tol.jsonl
      RC_TOL_REPORT=$OUT/tol.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider > $OUT/pytest_tol.log 2>&1
      tail -3 $OUT/pytest_tol.log
      python tools/tolerance_report.py $OUT/tol.jsonl > $OUT/tolerances.txt 2>&1
      head -5 $OUT/tolerances.txt
      gzip -f $OUT/tol.jsonl
      ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
      tail -2 $OUT/smoke.log
      ;;
    bench)
      name=${rest%%:*}; args=${rest#*:}
      [ -z "$rest" ] && name=bench && args=""
      [ "$args" == "$rest" ] && args=""
      t0=$(date +%s.%N)
      timeout 900 python bench.py $args > $OUT/$name.json 2> $OUT/$name.err
      echo "$name: $(python -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s wall" | tee $OUT/$name.wall.txt
      summ $OUT/$name.json
      ;;
    ab)
      name=${rest%%:*}; r2=${rest#*:}; envs=${r2%%:*}; args=${r2#*:}
      [ "$args" == "$r2" ] && args=""
      ( for kv in ${envs//,/ }; do export "$kv"; done; timeout 600 python bench.py $args > $OUT/$name.json 2> $OUT/$name.err )
      summ $OUT/$name.json
      ;;
    prof)
      name=${rest%%:*}; args=${rest#*:}
      [ "$args" == "$rest" ] && args=""
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o kt --output-format csv -- \
          python $R/bench.py $args --no-cpu-baseline --no-roofline > $OUT/prof_$name.log 2>&1 )
      find $OUT/prof_$name -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
      f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && cut -c1-110 "$f" | head -14
      ;;
    profenv)   # prof with environment switches: profenv:<name>:<ENV=V,ENV=V>:<bench.py args>
      name=${rest%%:*}; r2=${rest#*:}; envs=${r2%%:*}; args=${r2#*:}
      [ "$args" == "$r2" ] && args=""
      ( for kv in ${envs//,/ }; do export "$kv"; done; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o kt --output-format csv -- \
          python $R/bench.py $args --no-cpu-baseline --no-roofline > $OUT/prof_$name.log 2>&1 )
      find $OUT/prof_$name -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
      f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && cut -c1-110 "$f" | head -14
      ;;
    pmc)   # FETCH_SIZE / WRITE_SIZE passes of a training step, calibrated on a table copy (tools/pmc_collect.sh, pmc_summarize.py);
           # pmc = the BPRMF step -> pmc/, pmc:<name>:<workload args> e.g. pmc:neumf:--workload neumf -> pmc_<name>/
      if [ -z "$rest" ]; then
        bash tools/pmc_collect.sh $TAG/pmc > $OUT/pmc.log 2>&1
        head -16 $OUT/pmc/pmc_summary.txt
      else
        name=${rest%%:*}; args=${rest#*:}
        bash tools/pmc_collect.sh $TAG/pmc_$name "$args" > $OUT/pmc_$name.log 2>&1
        head -16 $OUT/pmc_$name/pmc_summary.txt
      fi
      ;;
    py)
      name=${rest%%:*}; args=${rest#*:}
      timeout 900 python $args > $OUT/$name.log 2>&1
      tail -15 $OUT/$name.log
      ;;
    *) echo "unknown stage $st" ;;
  esac
done
ls -laR $OUT > $OUT/ls.txt 2>&1

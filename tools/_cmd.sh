cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r02e_prof_deepfm
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02e_prof_deepfm/prof -o kt --output-format csv -- python $R/bench.py --workload deepfm --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/r02e_prof_deepfm/prof.log 2>&1
cd $R
find gpurun_out/r02e_prof_deepfm -name "*kernel_trace.csv" -size +20M -delete
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/r02e_prof_deepfm/prof/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>5} {float(r["Percentage"]):6.2f}%  {r["Name"][:100]}')
PY

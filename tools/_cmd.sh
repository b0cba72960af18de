timeout 600 python -m pytest tests/test_gpu_plan.py tests/test_gpu_bprmf.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('ahead', round(j['ms_per_step'],4), round(j['value']/1e6,2), round(j['step_effective_gbps'],1), {k:round(v,4) for k,v in j['phases_ms'].items()}, j['roofline'].get('alone',{}).get('avg_ms'))"
done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_e_host_tail.py -x -q -k "runner_policy" -s 2>&1 | grep -E "parity|passed|failed|Error" | head
timeout 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/bench_b8.json 2> gpurun_out/bench_b8.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b8.json'))
print('B=8', round(d['value'],1), 'from_host', round(d['from_host']['value'],1), 'runner_policy', {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['runner_policy'].items() if k!='what'}, 'latency', round(d['latency_ms_per_step'],2))
PY
timeout 400 python bench.py --batch 1 --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b1.json'))
print('B=1', round(d['value'],1), 'from_host', round(d['from_host']['value'],1), 'runner_policy', {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['runner_policy'].items() if k!='what'}, 'latency', round(d['latency_ms_per_step'],2))
PY
timeout 200 python scripts/bench_stem.py 2>&1 | tail -6

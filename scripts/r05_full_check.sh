#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for b in 1 8; do
timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_b$b.json 2> gpurun_out/bench_b$b.log
python - $b <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/bench_b{sys.argv[1]}.json')); r=d['roofline']
print('B='+sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step; latency', round(d['latency_ms_per_step'],2), 'conv ms', round(r['all_conv_ms_per_step'],2))
PY
done

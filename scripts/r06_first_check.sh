#!/bin/bash
# round 6, first GPU contact: the new fail-safe + sharded tests, the stage goldens, one bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_g_persistent_rnn.py -x -q -m gpu -k "giveup or handoff or sticky" -s 2>&1 | tail -25
timeout 1200 python -m pytest tests/test_gpu_h_sharded.py -x -q -m gpu -s 2>&1 | tail -40
timeout 900 python -m pytest tests/test_gpu_a_stages.py tests/test_gpu_g_persistent_rnn.py -x -q -m gpu 2>&1 | tail -5
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_first.json 2> gpurun_out/bench_first.log
tail -3 gpurun_out/bench_first.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_first.json')); r=d['roofline']
print(round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step; latency', round(d['latency_ms_per_step'],2), 'frac', r['frac'])
PY

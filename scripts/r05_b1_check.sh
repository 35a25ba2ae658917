#!/bin/bash
# op tests of the new small-grid paths + one-image bench with the conv table + batch-8 A/B of the routing switches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f_ops.py -x -q -k "splitk or odd_width or ragged or winograd" 2>&1 | tail -6
timeout 300 python bench.py --batch 1 --steps 100 --warmup 10 --no-cpu-baseline --conv-table gpurun_out/conv_table_b1.txt > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b1.json')); r=d['roofline']
print('B=1', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step; latency', round(d['latency_ms_per_step'],2), 'conv ms', round(r['all_conv_ms_per_step'],2))
PY
bash scripts/ab_bench.sh "GLASS_SPLITK=0 GLASS_SMALL_GRID=0" "GLASS_SPLITK=1 GLASS_SMALL_GRID=1" 2>&1 | tee gpurun_out/ab_b8_small_grid.txt

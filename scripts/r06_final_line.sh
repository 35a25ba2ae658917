#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_f_ops.py -q -m gpu -k "splitk" -s 2>&1 | grep "default route\|passed\|failed"
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py --conv-table $O/conv_table.txt > $O/bench.json 2> $O/bench.log; tail -1 $O/bench.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json')); r=d['roofline']
print(round(d['value'],1), round(d['ms_per_step'],2), 'latency', round(d['latency_ms_per_step'],2), 'frac', round(r['frac'],3), 'sust', round(r['frac_of_sustained_peak'],3), 'traffic', r['traffic'], 'pmc', r['mfma_util_percent_pmc'], 'rocprof_avg_us', r['rocprof_avg_us'], 'handoff', d.get('recurrent_handoff_status'))
PY

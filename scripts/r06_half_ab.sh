#!/bin/bash
# F(4x4) half shape (16 tiles x 64 channels, two workgroups per CU) vs the narrow shape of rounds 2-5 on the 64-channel layers
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f_ops.py -x -q -m gpu -k "winograd43 or wino" 2>&1 | tail -3
for rep in 1 2; do
for v in 1 0; do
  echo "== GLASS_W43_NARROW=$v"; GLASS_W43_NARROW=$v W43_LAYERS=4,5,6,3 python scripts/bench_w43.py 2>&1 | grep ABL
done; done | tee gpurun_out/w43_half_layers.txt
GLASS_W43_NARROW=0 python scripts/exp_w43_accuracy.py 2>&1 | grep narrow
for i in 1 2 3; do
  for v in 1 0; do
    echo -n "[narrow=$v] "; GLASS_W43_NARROW=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"
  done
done | tee gpurun_out/w43_half_bench.txt

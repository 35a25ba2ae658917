#!/bin/bash
# one-image line twice with and without the dual-source blocks (lease check of the B=1 figures), then the bench under torchrun
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
for i in 1 2; do for v in 1 0; do
  echo -n "[B=1 dual=$v] "; GLASS_PW_DUAL=$v timeout 400 python bench.py --batch 1 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python scripts/ab_line.py
done; done | tee $O/b1_check.txt
timeout 400 python bench.py --batch 1 --steps 200 --warmup 10 --no-cpu-baseline --conv-table $O/conv_table_b1.txt > $O/bench_b1.json 2> $O/bench_b1.log
python -c "
import json; d=json.load(open('gpurun_out/bench_b1.json')); print('b1 line', round(d['value'],1), round(d['ms_per_step'],2), round(d['latency_ms_per_step'],2))"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 2>/dev/null | tail -1 | python scripts/ab_line.py

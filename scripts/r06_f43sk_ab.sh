#!/bin/bash
# F(4x4) split-K routing on / off, alternating: one image per step (latency is the third number) and the 8-image headline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3; do for v in 0 1; do
  echo -n "[B=1 GLASS_F43_SPLITK=$v] "; GLASS_F43_SPLITK=$v timeout 300 python bench.py --batch 1 --no-cpu-baseline --steps 200 --warmup 10 2>/dev/null | python scripts/ab_line.py
done; done | tee gpurun_out/f43_splitk_b1.txt
for i in 1 2 3; do for v in 0 1; do
  echo -n "[B=8 GLASS_F43_SPLITK=$v] "; GLASS_F43_SPLITK=$v timeout 300 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python scripts/ab_line.py
done; done | tee gpurun_out/f43_splitk_b8.txt

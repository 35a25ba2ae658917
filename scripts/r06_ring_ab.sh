#!/bin/bash
# F(4x4) weight ring depth: 4 slots (3 groups = 768 cycles ahead, the product) vs 6 (5 groups ahead)
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/glass-text-spotting_amd
for v in ring6 ring9; do echo "== $v"; GLASS_HIP_LIB=$R/libglass_hip_$v.so W43_LAYERS=0,1,2,3 python scripts/bench_w43.py 2>&1 | grep ABL; done
for i in 1 2 3; do for v in ring4 ring6 ring9; do
  echo -n "[$v] "; GLASS_HIP_LIB=$R/libglass_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python scripts/ab_line.py
done; done

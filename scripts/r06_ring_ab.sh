#!/bin/bash
# F(4x4) schedule variants (variant libraries of one source tree; $VARIANTS): phase stamps at a full grid, per-layer times, alternating end-to-end runs
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/glass-text-spotting_amd
V=${VARIANTS:-"nopf pf"}
timeout 900 python -m pytest tests/test_gpu_f_ops.py -q -m gpu -k "winograd or wino or f43" 2>&1 | tail -2
for v in $V; do echo "== $v"; GLASS_HIP_LIB=$R/libglass_hip_$v.so python scripts/exp_w43_epilogue.py 2>&1 | grep -A1 "workgroups" | grep "dbg\|==" | sed 's/ | block life.*//'
  GLASS_HIP_LIB=$R/libglass_hip_$v.so python scripts/bench_w43.py 2>&1 | grep ABL; done
for i in 1 2 3 4; do for v in $V; do
  echo -n "[$v] "; GLASS_HIP_LIB=$R/libglass_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python scripts/ab_line.py
done; done

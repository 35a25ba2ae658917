#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/glass-text-spotting_amd
export GLASS_F43_SPLITK=0
for i in 1 2 3 4; do for v in e391 main; do
  if [ $v = main ]; then unset GLASS_HIP_LIB; else export GLASS_HIP_LIB=$R/libglass_hip_$v.so; fi
  echo -n "[$v] "; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python scripts/ab_line.py
done; done
for v in e391 main; do
  if [ $v = main ]; then unset GLASS_HIP_LIB; else export GLASS_HIP_LIB=$R/libglass_hip_$v.so; fi
  echo "== $v"; W43_LAYERS=0,1,2 python scripts/bench_w43.py 2>&1 | grep ABL
done

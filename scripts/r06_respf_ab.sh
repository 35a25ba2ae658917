#!/bin/bash
# F(4x4): residual lines touched during the last k-tile (prefetch into L2 / Infinity Cache) vs not
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/glass-text-spotting_amd
timeout 600 python -m pytest tests/test_gpu_f_ops.py -x -q -m gpu -k "winograd43" 2>&1 | tail -2
for rep in 1 2; do for v in w43nopf main; do
  if [ $v = main ]; then unset GLASS_HIP_LIB; else export GLASS_HIP_LIB=$R/libglass_hip_$v.so; fi
  echo "== $v"; W43_RES=1 W43_LAYERS=0,1,3,4 python scripts/bench_w43.py 2>&1 | grep ABL
done; done
for i in 1 2 3 4; do for v in w43nopf main; do
  if [ $v = main ]; then unset GLASS_HIP_LIB; else export GLASS_HIP_LIB=$R/libglass_hip_$v.so; fi
  echo -n "[$v] "; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"
done; done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/glass-text-spotting_amd
timeout 900 python -m pytest tests/test_gpu_f_ops.py -x -q -m gpu -k "winograd43 or wino" 2>&1 | tail -3
for v in w43r5 main; do
  if [ $v = main ]; then unset GLASS_HIP_LIB; else export GLASS_HIP_LIB=$R/libglass_hip_$v.so; fi
  python scripts/exp_w43_accuracy.py 2>&1 | grep "max|err"
done | tee gpurun_out/w43_accuracy.txt
for rep in 1 2; do
for v in w43vacc main; do
  if [ $v = main ]; then unset GLASS_HIP_LIB; else export GLASS_HIP_LIB=$R/libglass_hip_$v.so; fi
  echo "== $v"; python scripts/bench_w43.py 2>&1 | grep ABL
done; done | tee gpurun_out/w43_ab2_layers.txt
unset GLASS_HIP_LIB
for i in 1 2 3; do
  for v in w43r5 w43vacc main; do
    if [ $v = main ]; then unset GLASS_HIP_LIB; else export GLASS_HIP_LIB=$R/libglass_hip_$v.so; fi
    echo -n "[$v] "; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"
  done
done | tee gpurun_out/w43_ab2_bench.txt
unset GLASS_HIP_LIB
timeout 1500 python -m pytest tests/test_gpu_a_stages.py tests/test_gpu_b_configs.py -x -q -m gpu 2>&1 | tail -3

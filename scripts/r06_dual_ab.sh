#!/bin/bash
# the shortcut-into-conv3 dual-source launch (GLASS_PW_DUAL): tests, then alternating end-to-end A/B at 8 images and at one image
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f_ops.py tests/test_gpu_a_stages.py -m gpu -x -q -s -k "dual or backbone or end_to_end" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/dual_tests.log
tail -5 gpurun_out/dual_tests.log
{
for i in 1 2 3 4 5; do for v in 0 1; do
  echo -n "[dual=$v B=8] "; GLASS_PW_DUAL=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python scripts/ab_line.py
done; done
for i in 1 2 3; do for v in 0 1; do
  echo -n "[dual=$v B=1] "; GLASS_PW_DUAL=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --batch 1 2>/dev/null | python scripts/ab_line.py
done; done
} | tee gpurun_out/dual_ab.txt

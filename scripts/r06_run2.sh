#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
./.scratch/clock_calib 2>&1 | tee gpurun_out/clock_calib.txt
GLASS_W43_ABL=4 GLASS_W43_DBG_DUMP=/tmp/w43.bin python scripts/exp_w43_gap.py 2>/tmp/w43_gap.err > gpurun_out/w43_gap.txt; tail -3 /tmp/w43_gap.err; cat gpurun_out/w43_gap.txt
timeout 1200 python -m pytest tests/test_gpu_b_configs.py -x -q -m gpu -s -k "fp16" 2>&1 | grep -v "^$" | grep "parity\|ulps\|teacher\|passed\|failed\|Error\|assert" | tee gpurun_out/fp16_tests.txt | tail -60

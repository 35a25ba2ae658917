#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
timeout 1700 python -m pytest tests/test_gpu_a_stages.py tests/test_gpu_b_configs.py tests/test_gpu_e_host_tail.py tests/test_gpu_f_ops.py tests/test_gpu_z_pipeline.py -x -q -m gpu 2>&1 | tail -4
timeout 400 python bench.py --batch 1 --steps 200 --warmup 10 --no-cpu-baseline --conv-table $O/conv_table_b1.txt > $O/bench_b1.json 2> $O/bench_b1.log
python scripts/ab_line.py < $O/bench_b1.json; head -40 $O/conv_table_b1.txt

#!/bin/bash
# Round-5 artefacts with the CURRENT library in one lease: kernel stats + PMC (fp32, fp16s), the bench lines, configs[4] counters.
cd "$GRAFT_REPO_ROOT"
bash scripts/collect_profiles.sh > gpurun_out/collect.log 2>&1
bash scripts/r05_final_lines.sh 2>&1 | tail -8
bash scripts/r05_textocr_counters.sh > gpurun_out/textocr.log 2>&1; tail -1 gpurun_out/textocr.log
GLASS_SINGLE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_b1 -o r -- python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 > gpurun_out/bench_b1_under_rocprof.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_b1 -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 (one image per step, one step at a time, one stream)" > gpurun_out/kernel_stats_b1.txt
head -4 gpurun_out/kernel_stats_b1.txt | cut -c1-200

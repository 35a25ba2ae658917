#!/bin/bash
# Run ON the GPU box (via gpurun): kernel trace + the three PMC passes of bench.py, summarised into gpurun_out/.
#   gpurun --timeout 1500 -- 'bash scripts/collect_profiles.sh'
# Afterwards, in the build container:  for f in kernel_stats_default kernel_stats_serial pmc_conv_summary ...; cp gpurun_out/$f profiles/rNN_$f
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# 1. kernel trace of the default bench (two steps in flight, two-stream local extractor, as shipped)
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_default -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/bench_under_rocprof.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_default -name '*.db' | head -1)" 0 "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras (default: 2 steps in flight, two-stream local extractor)" > gpurun_out/kernel_stats_default.txt
python scripts/prof_by_grid.py "$(find /tmp/pr_default -name '*.db' | head -1)" conv3x3_wino43_f32 "conv3x3_wino43_f32 per layer shape INSIDE the step (default bench: two steps in flight, two-stream local extractor)" > gpurun_out/wino43_by_shape_default.txt
# 2. the same, one step at a time on a single stream (per-kernel durations comparable with bench.py's serial metering step)
GLASS_SINGLE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_serial -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 > gpurun_out/bench_under_rocprof_serial.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_serial -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 (one step at a time, one stream)" > gpurun_out/kernel_stats_serial.txt
python scripts/prof_by_grid.py "$(find /tmp/pr_serial -name '*.db' | head -1)" conv3x3_wino43_f32 "conv3x3_wino43_f32 per layer shape on the SERIAL meter (one step at a time, one stream)" > gpurun_out/wino43_by_shape_serial.txt
# 3. PMC passes (counters only, own runs)
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  GLASS_SINGLE_STREAM=1 timeout 500 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$tag -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > gpurun_out/pmc_$tag.log 2>&1
  python scripts/pmc_summary.py "$(find /tmp/pmc_$tag -name '*.db' | head -1)" > gpurun_out/pmc_$tag.json
done
python scripts/pmc_make_summary.py gpurun_out/pmc_FETCH_SIZE.json gpurun_out/pmc_WRITE_SIZE.json gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES.json gpurun_out/kernel_stats_serial.txt > gpurun_out/pmc_conv_summary.json
# 3b. BASELINE configs[4]'s precision (fp16 storage): serial kernel trace + the same three PMC passes -> pmc_fp16s_summary.json
if [ "${GLASS_COLLECT_FP16S:-1}" = "1" ]; then
  GLASS_SINGLE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_fp16s -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 --precision fp16s > gpurun_out/bench_under_rocprof_fp16s.log 2>&1
  python scripts/prof_summary.py "$(find /tmp/pr_fp16s -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 --precision fp16s (one step at a time, one stream)" > gpurun_out/kernel_stats_fp16s_serial.txt
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    GLASS_SINGLE_STREAM=1 timeout 500 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc16_$tag -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 --precision fp16s > gpurun_out/pmc_fp16s_$tag.log 2>&1
    python scripts/pmc_summary.py "$(find /tmp/pmc16_$tag -name '*.db' | head -1)" > gpurun_out/pmc_fp16s_$tag.json
  done
  python scripts/pmc_make_summary.py --fp16s gpurun_out/pmc_fp16s_FETCH_SIZE.json gpurun_out/pmc_fp16s_WRITE_SIZE.json gpurun_out/pmc_fp16s_SQ_VALU_MFMA_BUSY_CYCLES.json gpurun_out/kernel_stats_fp16s_serial.txt > gpurun_out/pmc_fp16s_summary.json
  head -40 gpurun_out/pmc_fp16s_summary.json
fi
tail -1 gpurun_out/bench_under_rocprof.log | cut -c1-300
cat gpurun_out/wino43_by_shape_default.txt gpurun_out/wino43_by_shape_serial.txt
head -12 gpurun_out/kernel_stats_serial.txt | cut -c1-150
cat gpurun_out/pmc_conv_summary.json | head -60
# 4. optional: the N-rank bench line (one process per GPU over RCCL; on a 1-GPU box GLASS_BENCH_BACKEND=gloo puts every rank on
#    device 0 and still exercises the sharding + all-gather path):  GLASS_BENCH_BACKEND=gloo bash scripts/collect_profiles.sh 2
N=${1:-1}
if [ "$N" -gt 1 ]; then
  # the driver-facing entry itself: bench.py starts its N ranks (glass_amd.distributed.launch_local_ranks), no torchrun
  timeout 900 python bench.py --gpus "$N" --steps 20 --warmup 3 --no-extras > gpurun_out/bench_${N}rank.json 2> gpurun_out/bench_${N}rank.log
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_${N}rank.json')); print({k: d[k] for k in ('value','n_gpus','ms_per_step','comm')})"
fi

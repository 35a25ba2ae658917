#!/bin/bash
# BASELINE configs[4] at ITS OWN shape and precision (1333 long side -> 1344^2 padded, 100 RoIs, fp16 storage): bench line, conv table,
# serial kernel stats and the three PMC passes -> gpurun_out/*textocr*
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
A="--side 1333 --rois 100 --precision fp16s"
timeout 500 python bench.py $A --steps 20 --warmup 3 --no-cpu-baseline --conv-table $O/conv_table_textocr_fp16s.txt > $O/bench_textocr_fp16s.json 2> $O/bench_textocr_fp16s.log
GLASS_SINGLE_STREAM=1 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/pr_tx -o r -- python bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 > $O/bench_textocr_under_rocprof.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_tx -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1" > $O/kernel_stats_textocr_fp16s_serial.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  GLASS_SINGLE_STREAM=1 timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmctx_$tag -o pmc -- python bench.py $A --steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > $O/pmc_textocr_$tag.log 2>&1
  python scripts/pmc_summary.py "$(find /tmp/pmctx_$tag -name '*.db' | head -1)" > $O/pmc_textocr_fp16s_$tag.json
done
python scripts/pmc_make_summary.py --fp16s $O/pmc_textocr_fp16s_FETCH_SIZE.json $O/pmc_textocr_fp16s_WRITE_SIZE.json $O/pmc_textocr_fp16s_SQ_VALU_MFMA_BUSY_CYCLES.json $O/kernel_stats_textocr_fp16s_serial.txt > $O/pmc_textocr_fp16s_summary.json
head -50 $O/pmc_textocr_fp16s_summary.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_textocr_fp16s.json')); r=d['roofline']
print(round(d['value'],1),'img/s', round(d['ms_per_step'],2),'ms; dominant', r['kernel'][:40], 'frac', round(r['frac'],3), 'ms', round(r['kernel_ms_per_step'],2))
PY

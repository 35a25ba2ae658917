#!/bin/bash
# Round-5 first measurements (run ON the GPU box): single-image lines, BASELINE configs[4] at its own shape, power / clock log.
#   gpurun --timeout 1700 -- 'bash scripts/r05_measure_a.sh'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out
# --- single image (VERDICT r4 missing #2): throughput with 2 in flight + latency, 1000^2 / 32 RoIs and the runner-policy size
timeout 300 python bench.py --batch 1 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.log
timeout 300 python bench.py --batch 1 --side 1200 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_b1_1200.json 2> $O/bench_b1_1200.log
GLASS_SINGLE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_b1 -o r -- python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 > $O/bench_b1_under_rocprof.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_b1 -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 (one image per step, one step at a time, one stream)" > $O/kernel_stats_b1.txt

# --- BASELINE configs[4] at its own shape (1333 long side -> 1344^2 padded, 100 RoIs), fp16s (its precision) and fp32
timeout 500 python bench.py --side 1333 --rois 100 --precision fp16s --steps 20 --warmup 3 --no-cpu-baseline --conv-table $O/conv_table_textocr_fp16s.txt > $O/bench_textocr_fp16s.json 2> $O/bench_textocr_fp16s.log
timeout 500 python bench.py --side 1333 --rois 100 --steps 20 --warmup 3 --no-cpu-baseline --conv-table $O/conv_table_textocr_fp32.txt > $O/bench_textocr_fp32.json 2> $O/bench_textocr_fp32.log
GLASS_SINGLE_STREAM=1 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/pr_tx -o r -- python bench.py --side 1333 --rois 100 --precision fp16s --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 > $O/bench_textocr_under_rocprof.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_tx -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py --side 1333 --rois 100 --precision fp16s --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1" > $O/kernel_stats_textocr_fp16s_serial.txt
# --- power / clock log (VERDICT r4 #4b)
timeout 120 python scripts/power_clock_log.py 6 > $O/power_clock.txt 2>&1
for f in bench_b1 bench_b1_1200 bench_textocr_fp16s bench_textocr_fp32; do
  python - "$O/$f.json" <<'EOF'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], {k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("value", "ms_per_step", "latency_ms_per_step") if k in d},
          "from_host", round(d.get("from_host", {}).get("value", 0), 1), "roofline.frac", round(d["roofline"]["frac"], 3), d["roofline"]["kernel"][:30])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
EOF
done
head -30 $O/kernel_stats_b1.txt | cut -c1-160
grep -E "steady|==|path" $O/power_clock.txt

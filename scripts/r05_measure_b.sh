#!/bin/bash
# Round-5: per-layer conv table at one image per step + the power / clock log of the GPU the process computes on.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python bench.py --batch 1 --steps 100 --warmup 10 --no-cpu-baseline --conv-table $O/conv_table_b1.txt > $O/bench_b1.json 2> $O/bench_b1.log
for d in 1 2 3 4; do
  echo -n "batch 1, $d steps in flight: "
  timeout 300 python bench.py --batch 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --pipeline $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms/step')"
done | tee $O/b1_pipeline_depth.txt
timeout 200 python scripts/power_clock_log.py 6 > $O/power_clock.txt 2>&1
grep -E "steady|==|path|PCI" $O/power_clock.txt
sort -r $O/conv_table_b1.txt | head -70

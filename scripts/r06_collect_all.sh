#!/bin/bash
# Round-6 artefacts with the CURRENT library in one lease: GPU suite (+ measured parity deltas), kernel stats + PMC (fp32, fp16s),
# the bench lines, configs[4] counters, the one-image trace, RCCL world-1 + 2-rank gloo lines, pipeline-depth check.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
( time timeout 1700 python -m pytest tests -q -m gpu -s ) > $O/gputest_full.log 2>&1; tail -3 $O/gputest_full.log
grep "^\[parity\]\|^\[fp16s\|^\[post-processor\|passed\|failed" $O/gputest_full.log > $O/gputest_parity_deltas.log
bash scripts/collect_profiles.sh > $O/collect.log 2>&1
bash scripts/r05_final_lines.sh 2>&1 | tail -8
bash scripts/r05_textocr_counters.sh > $O/textocr.log 2>&1; tail -1 $O/textocr.log
GLASS_SINGLE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_b1 -o r -- python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 > $O/bench_b1_under_rocprof.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_b1 -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 (one image per step, one step at a time, one stream)" > $O/kernel_stats_b1.txt
head -4 $O/kernel_stats_b1.txt | cut -c1-200
GLASS_BENCH_RCCL_WORLD1=1 timeout 400 python bench.py --steps 50 --no-cpu-baseline --no-extras > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.log
GLASS_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-extras > $O/bench_2rank_gloo_on_1gpu.json 2> $O/bench_2rank.log
for d in 2 3; do echo -n "pipeline $d: "; timeout 300 python bench.py --pipeline $d --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; done | tee $O/pipeline_depth.txt
python - <<'PY'
import json
for f in ("bench_rccl_world1","bench_2rank_gloo_on_1gpu"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["value"],1), d["n_gpus"], d["comm"])
    except Exception as e: print(f,"FAILED",e)
PY
timeout 900 python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-extras > $O/soak_1000steps_fp32.json 2> $O/soak.log
python -c "import json; d=json.load(open('gpurun_out/soak_1000steps_fp32.json')); print('soak', round(d['value'],1), d.get('recurrence_handoff_status'), d.get('hbm_peak_reserved_gb'))"

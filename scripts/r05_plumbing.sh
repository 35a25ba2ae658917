cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e_host_tail.py tests/test_gpu_z_pipeline.py -x -q 2>&1 | tail -3
GLASS_SINGLE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_serial -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 > gpurun_out/bench_under_rocprof_serial.log 2>&1
python scripts/prof_summary.py "$(find /tmp/pr_serial -name '*.db' | head -1)" 0 "GLASS_SINGLE_STREAM=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 (one step at a time, one stream)" > gpurun_out/kernel_stats_serial.txt
head -4 gpurun_out/kernel_stats_serial.txt | cut -c1-200
grep -E "at::native|rocclr|Cat" gpurun_out/kernel_stats_serial.txt | cut -c1-150
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; done

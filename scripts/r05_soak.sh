cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/soak_1000steps_fp32.json 2> gpurun_out/soak.log; python -c "
import json; d=json.load(open('gpurun_out/soak_1000steps_fp32.json')); print('fp32 B=8 1000 steps', round(d['value'],1), d['recurrent_handoff_status'], round(d['hbm_peak_reserved_gb'],1))"
timeout 600 python bench.py --batch 1 --pipeline 4 --steps 3000 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/soak_b1.json 2>> gpurun_out/soak.log; python -c "
import json; d=json.load(open('gpurun_out/soak_b1.json')); print('fp32 B=1 depth 4, 3000 steps', round(d['value'],1), d['recurrent_handoff_status'])"
timeout 600 python bench.py --precision fp16s --steps 1000 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/soak_fp16s.json 2>> gpurun_out/soak.log; python -c "
import json; d=json.load(open('gpurun_out/soak_fp16s.json')); print('fp16s B=8 1000 steps', round(d['value'],1), d['recurrent_handoff_status'])"
GLASS_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 3 --no-extras > gpurun_out/bench_2rank_gloo_on_1gpu.json 2>> gpurun_out/soak.log; python -c "
import json; d=json.load(open('gpurun_out/bench_2rank_gloo_on_1gpu.json')); print('2 ranks gloo on 1 GPU', round(d['value'],1), d['n_gpus'], d['comm']['gathered_records_shape'], d.get('cpu_baseline_ref',{}).get('source'), d['recurrent_handoff_status'])"
grep -E "preflight|bench rank" gpurun_out/soak.log | cut -c1-260
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

cd $GRAFT_REPO_ROOT
bash scripts/ab_bench.sh "GLASS_RNN=steps" "GLASS_RNN=1x1" 2>&1 | tee gpurun_out/ab_rnn.txt
for m in steps 1x1; do echo -n "[B=1 $m] "; GLASS_RNN=$m timeout 300 python bench.py --batch 1 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), 'latency', round(d['latency_ms_per_step'],2))"; done | tee -a gpurun_out/ab_rnn.txt
for m in steps 1x1 steps 1x1; do echo -n "[B=8 latency $m] "; GLASS_RNN=$m timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --pipeline 1 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; done | tee -a gpurun_out/ab_rnn.txt

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_f_ops.py -q -k "winograd" 2>&1 | tail -2
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/ws_n -o pmc -- python scripts/exp_write_size.py > gpurun_out/ws_n.log 2>&1
python scripts/pmc_summary.py "$(find /tmp/ws_n -name '*.db' | head -1)" > gpurun_out/ws_n.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/ws_n.json'))
for r in d.get('per_kernel',[])[:4]:
    print('WRITE_SIZE KB/launch', f"{r['mean_per_dispatch']:.5g}", '= x%.3f of the output' % (r['mean_per_dispatch']/524288), r['kernel'][:60])
PY
timeout 200 python scripts/exp_small_grid.py 2>&1 | sed -n 3,9p

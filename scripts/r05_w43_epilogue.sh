cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OLD=$PWD/glass-text-spotting_amd/libglass_hip_w43old.so
for lib in "" "$OLD"; do
  tag=$([ -z "$lib" ] && echo new || echo old)
  GLASS_HIP_LIB=$lib timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/ws_$tag -o pmc -- python scripts/exp_write_size.py > gpurun_out/ws_$tag.log 2>&1
  python scripts/pmc_summary.py "$(find /tmp/ws_$tag -name '*.db' | head -1)" > gpurun_out/ws_$tag.json
  python - $tag <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/ws_{sys.argv[1]}.json'))
for r in d.get('per_kernel',[])[:3]:
    print(sys.argv[1], 'WRITE_SIZE KB/launch', f"{r['mean_per_dispatch']:.5g}", '= x%.3f of the output' % (r['mean_per_dispatch']/524288), r['kernel'][:60])
PY
  GLASS_HIP_LIB=$lib timeout 200 python scripts/bench_w43.py 2>&1 | tail -8
done
for i in 1 2 3; do for lib in "" "$OLD"; do tag=$([ -z "$lib" ] && echo new || echo old); echo -n "[$tag] "; GLASS_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; done; done

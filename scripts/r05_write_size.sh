cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/ws_$c -o pmc -- python scripts/exp_write_size.py > gpurun_out/ws_$c.log 2>&1
  python scripts/pmc_summary.py "$(find /tmp/ws_$c -name '*.db' | head -1)" > gpurun_out/ws_$c.json
  python - $c <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/ws_{sys.argv[1]}.json'))
for r in d.get('per_kernel',[])[:14]:
    print(sys.argv[1], r['samples'], f"{r['mean_per_dispatch']:.4g}", r['kernel'][:80])
PY
done

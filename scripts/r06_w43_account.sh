#!/bin/bash
# VERDICT r5 #3 (first half): account for the F(4x4) kernel's gap between the stamp model (0.67) and the measured 0.55
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== clock calibration"; ./.scratch/clock_calib 2>&1 | tee gpurun_out/clock_calib.txt
echo "== product kernel, per shape"; python scripts/bench_w43.py 2>&1 | grep ABL | tee gpurun_out/w43_layers.txt
echo "== gap accounting"
GLASS_W43_ABL=4 GLASS_W43_DBG_DUMP=/tmp/w43.bin W43_SHADER_MHZ=${W43_SHADER_MHZ:-0} python scripts/exp_w43_gap.py 2>/tmp/w43_gap.err > gpurun_out/w43_gap.txt; tail -3 /tmp/w43_gap.err; cat gpurun_out/w43_gap.txt
echo "== counters"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u > gpurun_out/pmc_counter_names.txt; wc -l gpurun_out/pmc_counter_names.txt
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for L in 0 5; do
    W43_LAYERS=$L timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_w43_${i}_$L -o pmc -- python scripts/bench_w43.py > gpurun_out/pmc_w43_${i}_$L.log 2>&1
    python scripts/pmc_summary.py "$(find /tmp/pmc_w43_${i}_$L -name '*.db' | head -1)" > gpurun_out/pmc_w43_${i}_layer$L.json 2>>gpurun_out/pmc_w43_${i}_$L.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/pmc_w43_*_layer*.json')):
    d=json.load(open(f))
    print(f)
    for r in d.get('per_kernel',[]):
        if 'wino43' in r['kernel']: print('   ', r['counter'], r['samples'], round(r['mean_per_dispatch'],1), round(r['sum_duration_ns']/max(r['samples'],1)/1e3,1),'us')
    if 'error' in d: print(d['error'])
PY

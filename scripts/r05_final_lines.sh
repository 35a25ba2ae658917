cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python bench.py --conv-table $O/conv_table.txt > $O/bench.json 2> $O/bench.log
timeout 400 python bench.py --batch 1 --steps 200 --warmup 10 --no-cpu-baseline --conv-table $O/conv_table_b1.txt > $O/bench_b1.json 2> $O/bench_b1.log
timeout 400 python bench.py --batch 1 --side 1200 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_b1_1200.json 2> $O/bench_b1_1200.log
timeout 400 python bench.py --workload backbone --steps 50 --no-cpu-baseline --no-extras > $O/bench_backbone.json 2> $O/bench_backbone.log
timeout 400 python bench.py --precision fp16s --steps 50 --no-cpu-baseline > $O/bench_fp16_storage_mode.json 2> $O/bench_fp16s.log
timeout 400 python bench.py --precision fp16 --steps 50 --no-cpu-baseline > $O/bench_fp16_mode.json 2> $O/bench_fp16.log
timeout 500 python bench.py --side 1333 --rois 100 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_textocr_fp32.json 2> $O/bench_textocr_fp32.log
python - <<'PY'
import json
for f in ("bench","bench_b1","bench_b1_1200","bench_backbone","bench_fp16_storage_mode","bench_fp16_mode","bench_textocr_fp32"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); r=d["roofline"]
        print(f, round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "latency", round(d.get("latency_ms_per_step",0),2), "from_host", round(d.get("from_host",{}).get("value",0),1),
              "policy", round(d.get("runner_policy",{}).get("value",0),1), "| roofline", r["kernel"][:22], "frac", round(r["frac"],3), "traffic", r["traffic"], "pmc", r["mfma_util_percent_pmc"], "cpu", d.get("cpu_baseline",{}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY

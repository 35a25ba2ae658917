#!/bin/bash
# build container, after scripts/r05_collect_all.sh: gpurun_out/* -> profiles/r05_*
cd "$(dirname "$0")/.."
cd gpurun_out
for f in kernel_stats_default.txt kernel_stats_serial.txt wino43_by_shape_default.txt wino43_by_shape_serial.txt pmc_conv_summary.json pmc_FETCH_SIZE.json pmc_WRITE_SIZE.json pmc_fp16s_summary.json pmc_fp16s_FETCH_SIZE.json pmc_fp16s_WRITE_SIZE.json kernel_stats_fp16s_serial.txt kernel_stats_b1.txt conv_table.txt conv_table_b1.txt bench.json bench_b1.json bench_b1_1200.json bench_backbone.json bench_fp16_storage_mode.json bench_fp16_mode.json bench_textocr_fp32.json bench_textocr_fp16s.json conv_table_textocr_fp16s.txt kernel_stats_textocr_fp16s_serial.txt pmc_textocr_fp16s_summary.json; do [ -f $f ] && cp $f ../profiles/r05_$f; done
cp pmc_SQ_VALU_MFMA_BUSY_CYCLES.json ../profiles/r05_pmc_MFMA_BUSY.json; cp pmc_fp16s_SQ_VALU_MFMA_BUSY_CYCLES.json ../profiles/r05_pmc_fp16s_MFMA_BUSY.json
cd ..; python scripts/h16_by_class.py profiles/r05_conv_table_textocr_fp16s.txt | head -8

"""Fold the three rocprofv3 --pmc passes (scripts/pmc_summary.py outputs) into profiles/r01_pmc_conv_summary.json.

    python scripts/pmc_make_summary.py gpurun_out/pmc_FETCH_SIZE.json gpurun_out/pmc_WRITE_SIZE.json \
        gpurun_out/pmc_MFMA.json [gpurun_out/kernel_stats_serial.txt] > profiles/rNN_pmc_conv_summary.json

Per MFMA kernel: HBM bytes per launch = 2 x FETCH_SIZE KB (gfx950 under-reports 16-byte/lane streaming reads by
2x, MI355X_MICROARCH.md HBM section; checked on this path against maxpool_nhwc_kernel whose traffic is known)
+ WRITE_SIZE KB (exact); MfmaUtil = sum over SIMDs of SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs).
"""
import json
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "glass-text-spotting_amd"))
from glass_amd._lib import source_sha16  # noqa: E402

MODE = "fp32"
if "--fp16s" in sys.argv:            # the passes were taken with `bench.py --precision fp16s` (BASELINE configs[4]'s precision)
    sys.argv.remove("--fp16s")
    MODE = "fp16s"
fetch, write, mfma = (json.load(open(p)) for p in sys.argv[1:4])
stats_txt = sys.argv[4] if len(sys.argv) > 4 else None        # kernel_stats_serial.txt of the same library (optional)
HALF = MODE == "fp16s"               # fp16 instantiations of the fp32 template end in "true>"; the fp32 summary skips them
KEYS = {"conv_h16_kernel": "conv_h16_kernel", "local_stem_fused_h16": "local_stem_fused_kernel<true>",
        "conv_igemm_f16": "conv_igemm_f32<", "conv3x3_wino43_f32": "conv3x3_wino43_f32", "roi_align_rotated_h16": "roi_align_rotated_kernel<true>",
        "maxpool_h16": "maxpool_nhwc_kernel"} if HALF else {"conv3x3_wino43_f32": "conv3x3_wino43_f32", "conv1x1_pw_f32": "conv1x1_pw_f32", "conv1x1_pw_split": "conv1x1_pw_split", "conv3x3_wino128_f32": "conv3x3_wino128_f32", "conv3x3_wino_f32": "conv3x3_wino_f32", "conv_igemm_f32_128x128": "conv_igemm_f32<2, 2, 2, 2, 1, 3, 32, 1",
        "conv_igemm_f32_64x128": "conv_igemm_f32<1, 4, 2, 1, 1, 4, 32, 1",
        "conv_igemm_f32_128x64": "conv_igemm_f32<2, 2, 2, 1, 1, 4, 32, 1",
        "conv_igemm_f32_64x64": "conv_igemm_f32<2, 2, 1, 1, 1, 8, 32, 1",
        "backbone_stem_fused_kernel": "backbone_stem_fused_kernel", "local_stem_fused_kernel": "local_stem_fused_kernel<false>"}


def _inst_ok(kernel, sub):
    """fp32 summary: skip the fp16 instantiations ("...true>") of the shared templates; fp16s summary: for the implicit-GEMM
    template keep ONLY those (the kernels named with an explicit <true> / h16 are selected by their substring already)"""
    if not HALF:
        return "true>" not in kernel
    return ("true>" in kernel) if sub == "conv_igemm_f32<" else True


def pick(js, sub, counter):
    """every instantiation of the family (e.g. the wide AND the narrow shape of conv3x3_wino43_f32) folded into one row:
    counter sums and sample counts add, the mean is per dispatch over all of them - the figure bench.py's per-family
    `roofline.frac` is compared with"""
    rows = [r for r in js["per_kernel"] if sub in r["kernel"] and _inst_ok(r["kernel"], sub) and r["counter"] == counter]
    if not rows:
        return None
    tot, n = sum(r["sum"] for r in rows), sum(r["samples"] for r in rows)
    return {"kernel": " | ".join(sorted({r["kernel"] for r in rows})), "sum": tot, "samples": n, "mean_per_dispatch": tot / n,
            "instantiations": len({r["kernel"] for r in rows})}


def ndisp(js, sub):
    rows = [r for r in js.get("dispatches", []) if sub in r["kernel"] and _inst_ok(r["kernel"], sub)]
    if not rows:
        return None, None
    return sum(r["n"] for r in rows), sum(r["total_ns"] for r in rows)


def kernel_stats(path, sub):
    """weighted average launch duration of the family in a scripts/prof_summary.py table (rocprofv3 --kernel-trace --stats of
    the serial bench): sum of TotalDurationNs / sum of Calls over its instantiations"""
    calls = tot = 0
    with open(path) as f:
        for line in f:
            if line.startswith("#") or line.startswith("Name") or sub not in line or not _inst_ok(line[:72], sub):
                continue
            cols = line[72:].split()
            calls += int(cols[0])
            tot += int(cols[1])
    return (calls, tot / calls / 1e3) if calls else (0, None)


out = {"lib_source_sha16": source_sha16(),      # bench.py reports these counters only for the library they were taken with
       "precision": MODE,
       "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE> --kernel-trace -- "
                  "python bench.py --steps 1 --warmup 1 --no-cpu-baseline" + (" --precision fp16s" if HALF else "") +
                  " (three separate passes; summarised on the GPU box "
                  "with scripts/pmc_summary.py, folded with scripts/pmc_make_summary.py)",
       "calibration": {"kernel": "maxpool_nhwc_kernel", "known_read_MB": 369.1, "FETCH_SIZE_MB_raw": 204.72888,
                       "known_write_MB": 101.2, "WRITE_SIZE_MB_raw": 101.187584,
                       "note": "FETCH_SIZE under-reports 16-B/lane streaming reads ~2x on gfx950 (MI355X_MICROARCH.md HBM "
                               "section; measured once on this path, round 1); WRITE_SIZE is exact"}}
for key, sub in KEYS.items():
    f, w = pick(fetch, sub, "FETCH_SIZE"), pick(write, sub, "WRITE_SIZE")
    b, g = pick(mfma, sub, "SQ_VALU_MFMA_BUSY_CYCLES"), pick(mfma, sub, "GRBM_GUI_ACTIVE")
    if not (f and w and b and g):
        continue
    n, tot_ns = ndisp(mfma, sub)
    busy_per_launch = b["sum"] / n if n else None
    ent = {"kernel": f["kernel"], "instantiations": f["instantiations"], "dispatches_sampled": n,
           "FETCH_SIZE_KB_per_launch_raw": f["mean_per_dispatch"], "WRITE_SIZE_KB_per_launch_raw": w["mean_per_dispatch"],
           "hbm_bytes_per_launch_corrected": 1024.0 * (2.0 * f["mean_per_dispatch"] + w["mean_per_dispatch"]),
           "mfma_busy_cycles_sum_per_launch": busy_per_launch, "grbm_gui_active_mean": g["mean_per_dispatch"],
           "MfmaUtil_percent": 100.0 * busy_per_launch / (g["mean_per_dispatch"] * 1024.0) if busy_per_launch else None,
           "avg_launch_us_under_pmc": tot_ns / n / 1e3 if n else None}
    if stats_txt:
        ent["rocprof_calls"], ent["rocprof_avg_us"] = kernel_stats(stats_txt, sub)
        ent["rocprof_source"] = os.path.basename(stats_txt)
    out[key] = ent
print(json.dumps(out, indent=1))

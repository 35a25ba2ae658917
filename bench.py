#!/usr/bin/env python
"""bench.py — GLASS inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (`GlassRCNN.inference`: preprocess -> ResNet-50+FPN ->
rotated RPN -> box head -> rotated RoIAlign -> local extractor -> fusion attention -> recognizer
-> meta-arch postprocess) followed by the word post-processor (PostProcessorAcademic: merge,
thresholds, polygons, text decode) over one batch of 8 synthetic 1000x1000 images per GPU with 32 word RoIs per image
(BASELINE.json configs[2] = the configuration the metric "images/sec/GPU end-to-end spotting,
1000x1000, ~32 RoIs" is quoted on).  Inputs (float CHW images, injected word boxes) are resident
in HBM before the timed region.  Because random-init weights do not yield ~32 sensible word
detections, the RPN and box head run on their real 100 proposals and the recognition branch then
runs on 32 injected boxes per image — no stage is skipped.
With N > 1 every rank runs its own shard of images (weak scaling) and the per-image result records
are exchanged by ONE RCCL all_gather per step, inside the timed region.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL across processes needs this (the boxes export it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

BATCH = 8
SIDE = 1000
ROIS = 32
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FP16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (only used by --precision fp16)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gc-freeze", type=int, default=1, help="gc.freeze() the warm heap after warmup (0 = off)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--side", type=int, default=SIDE)
    ap.add_argument("--rois", type=int, default=ROIS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"],
                    help="fp32 (default: the reference's arithmetic, what the metric is quoted on) or fp16 = BASELINE configs[4]'s "
                         "precision: conv / linear operands rounded to fp16 on the fp16 matrix cores, fp32 accumulate and "
                         "storage.  An fp16 run is a separate, reduced-precision measurement, never the headline value.")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="steps in flight (host-side software pipelining over HIP streams, glass_amd/utils/pipeline.py): "
                         "while the host waits for one step's count read-back the other step's kernels keep the GPU busy "
                         "(1 = one step at a time: 204 vs 210 images/s on MI355X)")
    ap.add_argument("--workload", default="e2e", choices=["e2e", "backbone"],
                    help="e2e = BASELINE configs[2] (default, the metric's config); backbone = configs[1] (ResNet50-FPN only)")
    ap.add_argument("--conv-table", default="", help="write the per-launch conv table of the instrumented step here")
    ap.add_argument("--cpu-side", type=int, default=SIDE, help="image side of the bounded CPU-baseline sample")
    return ap.parse_args()


class ConvMeter:
    """Wraps the conv launches (glass_conv2d_nhwc / glass_conv3x3_winograd_nhwc) with HIP events on the launch
    stream (torch's current stream = the stream every kernel of the path is enqueued on) and tallies, per
    kernel family, the ALGORITHMIC FLOPs (direct-convolution count) and the FLOPs the
    kernel actually issues to the matrix cores (Winograd F(2x2,3x3): 16 instead of 36 MACs per 2x2 tile).
    Several identical steps are metered (`new_step()` between them) and every launch is charged the MEDIAN of its
    durations over the steps, so that one host hiccup during a metered step cannot distort the per-kernel figures."""

    def __init__(self, K):
        self.K = K
        self.orig = K.conv2d_nhwc
        self.steps = [[]]            # per metered step: [(family, x shape, w dims, stride, +res, algo, exec, e0, e1)]

    def new_step(self):
        self.steps.append([])

    def __enter__(self):
        def wrapped(x, w, bias=None, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.orig(x, w, bias, **kw)
            e1.record()
            cout, kh, kw_, cin = w.shape
            cin_real = 3 if cin == 4 else cin           # NHWC4-padded RGB inputs
            algo = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * kh * kw_ * cin_real
            path = self.K.last_conv_path()
            if path.startswith("winograd"):             # 16 MACs per (ceil(H/2) x ceil(W/2)) tile, channel pair
                ex = 2.0 * y.shape[0] * ((y.shape[1] + 1) // 2) * ((y.shape[2] + 1) // 2) * 16 * cout * cin
            else:
                ex = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * kh * kw_ * cin
            self.steps[-1].append((path, tuple(x.shape), (cout, kh, kw_), kw.get("stride", 1),
                                   kw.get("residual") is not None, algo, ex, e0, e1))
            return y
        self.K.conv2d_nhwc = wrapped
        return self

    def __exit__(self, *a):
        self.K.conv2d_nhwc = self.orig

    def _launches(self):
        """[(family, x shape, w dims, stride, +res, algo, exec, median ms)] in launch order"""
        torch.cuda.synchronize()
        steps = [s for s in self.steps if s]
        n = len(steps[0])
        assert all(len(s) == n and [r[:5] for r in s] == [r[:5] for r in steps[0]] for s in steps), \
            "metered steps must issue the same conv launches"
        rows = []
        for i in range(n):
            ms = sorted(s[i][7].elapsed_time(s[i][8]) for s in steps)
            rows.append(steps[0][i][:7] + (ms[len(ms) // 2],))
        return rows

    def table(self):
        """per-launch rows: path, input shape, (Cout, KH, KW), stride, +residual, ms, TFLOP/s (slowest first)"""
        return sorted(((p, xs, ws, st, res, ms, algo / ms / 1e9) for p, xs, ws, st, res, algo, _ex, ms in self._launches()),
                      key=lambda r: -r[5])

    def summary(self):
        out = {k: {"launches": 0, "ms": 0.0, "algo_flops": 0.0, "exec_flops": 0.0}
               for k in ("winograd128", "winograd", "direct", "direct_fp16")}
        for p, _xs, _ws, _st, _res, algo, ex, ms in self._launches():
            f = out[p]
            f["launches"] += 1
            f["ms"] += ms
            f["algo_flops"] += algo
            f["exec_flops"] += ex
        return out


def cpu_baseline(cfg, sd, side, rois):
    """Bounded CPU sample: the oracle (CPU restatement, kind='port') on ONE image of the workload."""
    from glass_amd.utils.synth import make_boxes, make_image
    from glass_amd.utils.host import usable_cpus
    from oracle import glass_cpu as O
    n = usable_cpus()                                             # affinity mask and cgroup CFS quota
    torch.set_num_threads(n)
    img = make_image(0, side, side).permute(2, 0, 1).float()
    boxes = [make_boxes(0, rois, side, side)]
    t0 = time.perf_counter()
    with torch.no_grad():
        O.glass_inference(sd, [img], cfg, injected_boxes=boxes)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "images/sec", "cores": n, "kind": "port",
            "sample": f"1 image {side}x{side}, {rois} injected RoIs, oracle/glass_cpu.py fp32 torch-CPU ({n} threads), "
                      f"single cold run {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("GLASS_BENCH_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" only to
    dev_index = local_rank % ndev                                 # exercise the N>1 plumbing on a 1-GPU box
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import glass_amd
    from glass_amd.utils.host import limit_host_threads
    limit_host_threads()          # before the first CPU tensor op: the host side is launch glue (utils/host.py)
    from glass_amd.config import get_glass_cfg
    from glass_amd.distributed import all_gather_records, pack_words
    from glass_amd.postprocess import build_post_processor
    from glass_amd.ops import native as K
    from glass_amd.utils.pipeline import drive, run_pipelined
    from glass_amd.utils.synth import make_boxes, make_image, make_state_dict

    cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"),
                        ["MODEL.DEVICE", f"cuda:{dev_index}", "MODEL.CONV_PRECISION", args.precision])
    sd = make_state_dict(1234)
    model = glass_amd.build_model(cfg)
    model.load_state_dict(sd)

    B = args.batch
    gidx = [rank * B + i for i in range(B)]                      # global image indices of this rank's shard
    images = [make_image(g, args.side, args.side).permute(2, 0, 1).float().contiguous().to(dev) for g in gidx]
    boxes = [make_boxes(g, args.rois, args.side, args.side).to(dev) for g in gidx]
    inputs = [{"image": im} for im in images]
    max_det = cfg.TEST.DETECTIONS_PER_IMAGE
    post = build_post_processor(cfg)                              # PostProcessorAcademic (device kernel)
    out_sizes = [(args.side, args.side)] * B
    steps_txt = cfg.MODEL.ROI_RECOGNIZER_HEAD.MAX_WORD_LENGTH + 1

    if args.workload == "backbone":
        il = model.preprocess_image(inputs)

    def local_step_g():
        """one step as a generator (glass_amd/utils/pipeline.py): yields where the host reads counts back"""
        if args.workload == "backbone":                           # BASELINE configs[1]: trunk + FPN only
            model.backbone.forward_nhwc(il.nhwc4)
            return torch.zeros((B, 1), device=dev)
            yield                                                 # pragma: no cover (makes this a generator)
        out = yield from model.inference_g(inputs, override_boxes=boxes)   # list[{"instances": Instances}] (views)
        det = out.batch                                           # + this step's padded device-resident batch
        # word post-processing (merge, thresholds, polygons, text decode + text-score filter) for the 8 images
        words = yield from post.process_padded_g(det.boxes, det.scores, det.counts_dev, det.text, None, out_sizes,
                                                 {"orientations": det.orient})
        return pack_words(words.words, max_det, steps_txt)        # fixed-size per-image word records

    def step_g():
        rec = yield from local_step_g()
        if dist is not None and backend != "nccl":
            return all_gather_records(rec.cpu())
        return all_gather_records(rec)

    def local_step():
        return drive(local_step_g())

    def run_steps(n):
        """n steps, `--pipeline` of them in flight (each on its own stream; host segments interleaved in a fixed
        order, so every rank issues its all_gathers in the same order)"""
        run_pipelined([step_g] * n, depth=args.pipeline, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # lazy initialisation, not part of W or K: the first step on each pipeline stream packs the Winograd weights and
    # fills that stream's allocator pool (hipMalloc); one step per stream so that a small --warmup cannot leave a cold one
    run_steps(args.pipeline)
    run_steps(args.warmup)
    if args.gc_freeze:
        # serving-loop hygiene, not skipped work: a generation-2 collection walks every object torch created at
        # import (~25 ms of host stall every few steps, visible once the step is host-bound); freeze moves the
        # warm heap to the permanent generation so later collections only look at per-step garbage
        import gc
        gc.collect()
        gc.freeze()
    barrier()
    prof = None
    if os.environ.get("GLASS_PROFILE_HOST"):           # diagnostic: cProfile of the timed loop's host side -> stderr
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(35)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    line = None
    if rank == 0:
        # dominant kernel: three extra instrumented steps, outside the timed region
        # (rank-local: must not enter a collective the other ranks are not in)
        # (serial execution: the product overlaps the two halves of the local extractor on two streams,
        #  which would inflate per-kernel event times)
        two = model.roi_heads.two_stream_local
        model.roi_heads.two_stream_local = False
        with ConvMeter(K) as meter:
            for rep in range(3):                                  # per-launch median over three identical steps
                if rep:
                    meter.new_step()
                local_step()
            fam = meter.summary()
            if args.conv_table:
                with open(args.conv_table, "w") as f:
                    for p_, xs, ws, st, res, ms, tf in meter.table():
                        f.write(f"{ms:8.3f} ms {tf:7.1f} TF/s  {p_:11s} x{list(xs)} -> Cout {ws[0]} k{ws[1]}x{ws[2]} s{st}"
                                f"{' +res' if res else ''}\n")
        model.roi_heads.two_stream_local = two
        conv_ms = sum(f["ms"] for f in fam.values())
        conv_flops = sum(f["algo_flops"] for f in fam.values())
        n_launch = sum(f["launches"] for f in fam.values())
        KNAME = {"winograd128": "conv3x3_wino128_f32 (Winograd F(2x2,3x3), fp32 MFMA, 32 tiles x 128 channels per workgroup)",
                 "winograd": "conv3x3_wino_f32 (Winograd F(2x2,3x3), fp32 MFMA, 64 tiles x 64 channels per workgroup)",
                 "direct": "conv_igemm_f32 (fp32 MFMA implicit-GEMM conv/linear)",
                 "direct_fp16": "conv_igemm_f32<..., HALF> (fp16 MFMA implicit-GEMM conv/linear, fp32 accumulate)"}
        PKEY = {"winograd128": "conv3x3_wino128_f32", "winograd": "conv3x3_wino_f32", "direct": "conv_igemm_f32_64x64",     # direct: its busiest instantiation
                "direct_fp16": "conv_igemm_f16"}
        PEAK = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else FP16_MFMA_PEAK_TFLOPS
        dom = max(fam, key=lambda k: fam[k]["ms"])            # the dominant kernel by time
        # PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate
        # rocprofv3 --pmc runs, gfx950 x2 read correction applied; scripts/pmc_make_summary.py)
        pmc_all = {}
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_conv_summary.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                pmc_all = json.load(f)

        def fam_entry(k):
            f = fam[k]
            if f["launches"] == 0:
                return None
            sec = f["ms"] * 1e-3
            pj = pmc_all.get(PKEY[k], {}) if isinstance(pmc_all.get(PKEY[k], {}), dict) else {}
            return {"kernel": KNAME[k], "launches_per_step": f["launches"], "kernel_ms_per_step": f["ms"],
                    "avg_launch_ms": f["ms"] / f["launches"],
                    "executed_tflops": f["exec_flops"] / sec / 1e12, "executed_frac": f["exec_flops"] / sec / 1e12 / PEAK,
                    "algorithmic_tflops": f["algo_flops"] / sec / 1e12,
                    "algorithmic_frac": f["algo_flops"] / sec / 1e12 / PEAK,
                    "algorithmic_gflop_per_launch": f["algo_flops"] / f["launches"] / 1e9,
                    "traffic": pj.get("hbm_bytes_per_launch_corrected"), "mfma_util_percent_pmc": pj.get("MfmaUtil_percent")}

        ent = fam_entry(dom)
        other = [fam_entry(k) for k in fam if k != dom and fam[k]["launches"]]
        line = {
            "metric": "images/sec/GPU end-to-end spotting, 1000x1000, ~32 RoIs; 1/2/4/8 GPU scaling",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16 operands, f32 accumulate/storage (reduced precision: not the headline)",
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: backbone + RotatedROIAlign + recognition head, "
                                    f"{args.rois} RoIs/img, bs={B}/GPU, {args.side}x{args.side} (padded to /32), fp32")
                       if args.workload == "e2e" else
                       f"BASELINE.json configs[1]: ResNet50-FPN backbone only, bs={B}/GPU, {args.side}x{args.side}, fp32",
                       "images_per_gpu_per_step": B, "rois_per_image": args.rois, "proposals_per_image": 100,
                       "weights": "random-init (seed 1234), reference architecture",
                       "parallelism": f"image-shard x{world}, 1 all_gather of result records/step",
                       "steps_in_flight": args.pipeline},
            "images_per_sec_per_gpu": value / world,
            "hbm_peak_reserved_gb": torch.cuda.max_memory_reserved(dev) / 1e9,   # rank 0, whole run (of 288 GB)
            "roofline": {"bound": "mfma", "kernel": ent["kernel"],
                         # `achieved` = ALGORITHMIC (direct-convolution) FLOP of the kernel's launches / their
                         # measured time, as the contract defines it; for the Winograd kernel this exceeds the MFMA peak
                         # (frac > 1) because it issues 2.25x fewer multiplies than the direct-convolution count.  The
                         # hardware-utilisation view - the FLOP the kernel really issues to the matrix cores - is
                         # `executed_tflops` / `executed_frac` (cross-checked by the PMC MFMA-busy counter).
                         "achieved": ent["algorithmic_tflops"], "peak": PEAK, "unit": "TFLOP/s",
                         "frac": ent["algorithmic_frac"],
                         "executed_tflops": ent["executed_tflops"], "executed_frac": ent["executed_frac"],
                         "traffic": ent["traffic"],
                         "traffic_note": "HBM bytes/launch of this kernel, PMC (profiles/r01_pmc_conv_summary.json)",
                         "mfma_util_percent_pmc": ent["mfma_util_percent_pmc"],
                         "launches_per_step": ent["launches_per_step"],
                         "algorithmic_gflop_per_launch": ent["algorithmic_gflop_per_launch"],
                         "avg_launch_ms": ent["avg_launch_ms"], "kernel_ms_per_step": ent["kernel_ms_per_step"],
                         "share_of_step": ent["kernel_ms_per_step"] / ms_per_step,
                         "other_mfma_kernels": other,
                         "all_conv_launches_per_step": n_launch, "all_conv_ms_per_step": conv_ms,
                         "all_conv_algorithmic_tflops": conv_flops / (conv_ms * 1e-3) / 1e12},
            "whole_step_tflops": conv_flops / (ms_per_step * 1e-3) / 1e12,
            # north star: throughput "as fraction of the conv-bound roofline" = fp32 MFMA peak / direct-conv FLOPs per
            # image (0.846 TFLOP at 1024^2, P=100, R=32 -> 186 img/s/GPU); the convs of this very step,
            # counted the same way, give the per-image figure used here.  > 1 is possible only because the 3x3 layers
            # run Winograd (2.25x fewer multiplies than the direct-convolution count).
            "conv_bound_roofline": {"tflop_per_image": conv_flops / B / 1e12,
                                    "images_per_sec_per_gpu": FP32_MFMA_PEAK_TFLOPS / (conv_flops / B / 1e12),
                                    "frac": (value / world) / (FP32_MFMA_PEAK_TFLOPS / (conv_flops / B / 1e12))},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, args.cpu_side, args.rois)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

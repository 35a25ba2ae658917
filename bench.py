#!/usr/bin/env python
"""bench.py — GLASS inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its N ranks itself (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # or joins the world torchrun made
It never runs fewer ranks than --gpus asks for: a node with fewer GPUs than N (RCCL backend) or a WORLD_SIZE that
contradicts --gpus exits non-zero without printing a line.

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
# what a PERFECT fp32-MFMA kernel sustains on this part: all 256 CUs issuing nothing but v_mfma_f32_16x16x4_f32 run at 2.10-2.20 GHz,
# not 2.40 (scripts/micro/clock_calib.hip -> profiles/r06_clock_calib.txt: 143.1 TFLOP/s; 32x32x2: 141.5).  `roofline.frac` keeps the
# nominal peak (the contract's definition); `frac_of_sustained_peak` is the same rate against this measured ceiling.
FP32_MFMA_SUSTAINED_TFLOPS = 143.1
DISTINCT_STEPS = 3                  # input sets cycled through the steps (images and boxes differ)
SURVEY_TFLOP_PER_IMAGE = 0.846    # direct-convolution work of the reference per 1000x1000 image, 100 proposals, 32 RoIs
FP16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (only used by --precision fp16)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200 = ~5.6 s of 8-image steps: long enough for an "
                                                              "outside clock / utilisation sampler to corroborate the line)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gc-freeze", type=int, default=1, help="gc.freeze() the warm heap after warmup (0 = off)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--side", type=int, default=SIDE)
    ap.add_argument("--rois", type=int, default=ROIS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "fp16s"],
                    help="fp32 (default: the reference's arithmetic, what the metric is quoted on) or fp16 = BASELINE configs[4]'s "
                         "precision: conv / linear operands rounded to fp16 on the fp16 matrix cores, fp32 accumulate and "
                         "storage; fp16s = the same arithmetic with the conv-path activations STORED as fp16 in HBM (fp16 storage).  "
                         "An fp16 / fp16s run is a separate, reduced-precision measurement, never the headline value.")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="steps in flight (host-side software pipelining over HIP streams, glass_amd/utils/pipeline.py): "
                         "while the host waits for one step's count read-back the other step's kernels keep the GPU busy "
                         "(1 = one step at a time: 204 vs 210 images/s on MI355X)")
    ap.add_argument("--workload", default="e2e", choices=["e2e", "backbone"],
                    help="e2e = BASELINE configs[2] (default, the metric's config); backbone = configs[1] (ResNet50-FPN only)")
    ap.add_argument("--conv-table", default="", help="write the per-launch conv table of the instrumented step here")
    ap.add_argument("--cpu-side", type=int, default=SIDE, help="image side of the bounded CPU-baseline sample")
    ap.add_argument("--from-host", action="store_true",
                    help="time ONLY the from-host variant as the main loop: every step starts from uint8 HWC images in pinned "
                         "host memory (H2D copy + on-device u8->f32 CHW conversion inside the step, GlassRunner's front-end). "
                         "Without this flag the contract's HBM-resident rate is `value` and the from-host rate of a second "
                         "timed loop is reported next to it (`from_host`).")
    ap.add_argument("--no-extras", action="store_true", help="skip the from-host and latency loops (value only)")
    ap.add_argument("--runner-policy", action="store_true",
                    help="time the GlassRunner path as the main loop: uint8 HWC host images -> H2D -> fused convert + bilinear resize "
                         "by the reference's policy (glass_runner.py:111-148: a 1000 x 1000 image under the ICDAR15 cfg's "
                         "MIN_SIZE_TEST 1200 is upscaled x1.2 to 1200 x 1200, padded to 1216 x 1216: 504 GMAC, BASELINE.md section 3) "
                         "-> model -> results un-scaled by 1 / 1.2.  Without the flag this leg is reported as `runner_policy` next "
                         "to `from_host`.  PCIe-inclusive and a larger workload than the metric's: never `value` of the headline.")
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
            if path == "winograd43r":                   # width 4 k + 1: F(4x4) on the k full tile columns + the last column direct (3 x 2 taps)
                ex = 2.0 * y.shape[0] * ((y.shape[1] + 3) // 4) * (y.shape[2] // 4) * 36 * cout * cin + \
                    2.0 * y.shape[0] * y.shape[1] * cout * 6 * cin
                path = "winograd43"
            elif path in ("winograd128r", "winogradr"):  # odd width: F(2x2) on the W // 2 full tile columns + the last column direct (3 x 2 taps)
                ex = 2.0 * y.shape[0] * ((y.shape[1] + 1) // 2) * (y.shape[2] // 2) * 16 * cout * cin + \
                    2.0 * y.shape[0] * y.shape[1] * cout * 6 * cin
                path = path[:-1]
            elif path == "winograd43k":                 # the same kernel as k-slices + an ordered reduction (one image in flight; small levels):
                ex = 2.0 * y.shape[0] * ((y.shape[1] + 3) // 4) * ((y.shape[2] + 3) // 4) * 36 * cout * cin      # same family, same executed count
                path = "winograd43"
            elif path == "winograd43":                  # F(4x4,3x3): 36 MACs per (ceil(H/4) x ceil(W/4)) tile, channel pair
                ex = 2.0 * y.shape[0] * ((y.shape[1] + 3) // 4) * ((y.shape[2] + 3) // 4) * 36 * cout * cin
            elif path == "pointwise_split":             # every fp32 product as 9 (or 6) exact bf16 piece products on the bf16 matrix cores
                ex = float(getattr(self.K.routing_of(w), "split", 0) or 9) * algo
            elif path.startswith("winograd"):           # 16 MACs per (ceil(H/2) x ceil(W/2)) tile, channel pair
                ex = 2.0 * y.shape[0] * ((y.shape[1] + 1) // 2) * ((y.shape[2] + 1) // 2) * 16 * cout * cin
            else:
                ex = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * kh * kw_ * cin
            self.steps[-1].append((path, tuple(x.shape), (cout, kh, kw_), kw.get("stride", 1),
                                   kw.get("residual") is not None, algo, ex, e0, e1))
            return y
        self.K.conv2d_nhwc = wrapped
        # the two fused stems (conv + conv/pool in one kernel) are convolution work too: metered as family "fused_stem" with the
        # direct-convolution FLOP of the convs they contain (max-pools are 0 FLOP)
        self.orig_bstem, self.orig_lstem = self.K.backbone_stem_fused, self.K.local_stem_fused

        def timed(fn, flops_of, tag):
            def run(x, *a, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = fn(x, *a, **kw)
                e1.record()
                fl = flops_of(x)
                self.steps[-1].append(("fused_stem", tuple(x.shape), tag, 1, False, fl, fl, e0, e1))
                return y
            return run
        self.K.backbone_stem_fused = timed(self.orig_bstem, lambda x: 2.0 * x.shape[0] * (x.shape[1] // 2) * (x.shape[2] // 2) * 64 * 49 * 3,
                                           (64, 7, 7))
        self.K.local_stem_fused = timed(self.orig_lstem, lambda x: 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * 9 * (16 * 3 + 32 * 16),
                                        (32, 3, 3))
        # a bottleneck block's shortcut + conv3 as one dual-source launch of the bf16-split kernel: the direct-convolution FLOP of BOTH convs
        self.orig_dual = self.K.conv1x1_dual_nhwc

        def dual(x1, x2, w, bias=None, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.orig_dual(x1, x2, w, bias, **kw)
            e1.record()
            algo = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3] * w.shape[3]
            self.steps[-1].append(("pointwise_split", tuple(x1.shape), (y.shape[3], 1, 1), kw.get("stride", 1), True, algo, 9.0 * algo, e0, e1))
            return y
        self.K.conv1x1_dual_nhwc = dual
        return self

    def __exit__(self, *a):
        self.K.conv2d_nhwc = self.orig
        self.K.backbone_stem_fused, self.K.local_stem_fused = self.orig_bstem, self.orig_lstem
        self.K.conv1x1_dual_nhwc = self.orig_dual

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
               for k in ("winograd43", "winograd128", "winograd", "pointwise", "pointwise_split", "direct", "direct_fp16", "packed_fp16", "fused_stem")}
        for p, _xs, _ws, _st, _res, algo, ex, ms in self._launches():
            f = out[p]
            f["launches"] += 1
            f["ms"] += ms
            f["algo_flops"] += algo
            f["exec_flops"] += ex
        return out


def cpu_baseline(cfg, sd, side, rois):
    """Bounded CPU sample: the oracle (CPU restatement, kind='port') on ONE image of the workload
    (config 3 shape at B = 1: side x side, `rois` injected RoIs) - 1 warm-up + 3 timed runs, median - plus the
    config-1 leg (one 512 x 512 image end to end, no injected boxes) and per-stage times of the last run."""
    import statistics
    from glass_amd.utils.synth import make_boxes, make_image
    from glass_amd.utils.host import usable_cpus
    from oracle import glass_cpu as O
    n = usable_cpus()                                             # affinity mask and cgroup CFS quota
    torch.set_num_threads(n)

    def timed(fn, warm=1, reps=3):
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            with torch.no_grad():
                fn()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return statistics.median(ts), ts

    img = make_image(0, side, side).permute(2, 0, 1).float()
    boxes = [make_boxes(0, rois, side, side)]
    med, ts = timed(lambda: O.glass_inference(sd, [img], cfg, injected_boxes=boxes))
    # per-stage split of one more run (same functions glass_inference calls, in its order)
    stages = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        x, sizes = O.preprocess([img], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 32)
        feats = O.resnet50_fpn(sd, x)
        t1 = time.perf_counter()
        props = O.rpn_proposals(sd, feats, sizes, cfg)
        t2 = time.perf_counter()
        pb = [p[0] for p in props]
        sc, dl, og, _ = O.box_head_logits(sd, feats, pb, cfg)
        O.box_inference(sc, dl, og, pb, sizes, cfg)
        t3 = time.perf_counter()
        O.recognizer_branch(sd, x, feats, boxes, cfg)
        t4 = time.perf_counter()
    stages = {"preprocess+backbone+fpn": (t1 - t0) * 1e3, "rpn": (t2 - t1) * 1e3, "box_head+nms": (t3 - t2) * 1e3,
              "recognition_branch": (t4 - t3) * 1e3}
    img1 = make_image(20, 512, 512).permute(2, 0, 1).float()
    med1, ts1 = timed(lambda: O.glass_inference(sd, [img1], cfg))
    return {"value": 1.0 / med, "unit": "images/sec", "cores": n, "kind": "port",
            "sample": f"1 image {side}x{side}, {rois} injected RoIs, oracle/glass_cpu.py fp32 torch-CPU ({n} threads), "
                      f"1 warm-up + 3 timed runs, median {med:.2f} s (runs {', '.join(f'{t:.2f}' for t in ts)} s)",
            "stage_ms": {k: round(v, 1) for k, v in stages.items()},
            "config1_512": {"value": 1.0 / med1, "unit": "images/sec",
                            "sample": f"BASELINE configs[0]: 1 image 512x512 end to end (no injected boxes), same protocol, "
                                      f"median {med1:.2f} s (runs {', '.join(f'{t:.2f}' for t in ts1)} s)"}}


def dry_run(args, rank, world, dist) -> None:
    """GLASS_BENCH_DRYRUN=1: the N-rank plumbing of this script WITHOUT a GPU or a model (the `-m "not gpu"` tests run it with
    world 2): the same launch path (self-launched ranks or torchrun), process group (gloo), `run_pipelined` schedule with a host
    read-back per step, ONE all_gather of fixed-size word records per step inside the timed region, max-over-ranks timing, `comm`
    block and line assertions as the real bench - only the step's records are synthetic.  Prints a line marked as a dry run; it is
    not a measurement."""
    from glass_amd.distributed import all_gather_records, words_record_size
    from glass_amd.utils.pipeline import ReadBack, run_pipelined
    B, max_det, steps_txt = args.batch, 100, 26
    width = words_record_size(max_det, steps_txt)

    def step_g(i):
        rec = torch.zeros((B, width))
        rec[:, 0] = float((rank * 7 + i) % 5)                       # word counts: differ by rank and step
        (counts,) = yield ReadBack(rec[:, 0].clone())                # a host read-back, like the real step's three
        assert counts.shape == (B,)
        return all_gather_records(rec, rows=B)

    def barrier():
        if dist is not None:
            dist.barrier()

    run_pipelined([(lambda i=i: step_g(i)) for i in range(args.warmup)], depth=args.pipeline, device="cpu")
    barrier()
    t0 = time.perf_counter()
    res = run_pipelined([(lambda i=i: step_g(100 + i)) for i in range(args.steps)], depth=args.pipeline, device="cpu")
    barrier()
    dt = time.perf_counter() - t0
    last = res[-1]
    assert last.dim() == 3 and tuple(last.shape) == (world, B, width), (tuple(last.shape), world, B, width)
    for r in range(world):                                          # every rank holds every rank's records of the LAST step
        assert float(last[r, 0, 0]) == float((r * 7 + 100 + args.steps - 1) % 5), (r, float(last[r, 0, 0]))
    comm = {"world_size": 1, "backend": None}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        per_rank = torch.empty((world,), dtype=torch.float64)
        dist.all_gather_into_tensor(per_rank, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                "per_rank_ms_per_step": [round(v / args.steps * 1e3, 3) for v in per_rank.tolist()],
                "gathered_records_shape": list(last.shape), "gathered_records_expected": world * B}
        dt = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line = {"metric": "DRY RUN of the N-rank plumbing (no GPU, no model): not a measurement", "value": world * B * args.steps / dt,
                "unit": "synthetic records/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt / args.steps * 1e3, "data": "dry-run", "comm": comm}
        if world > 1:
            line["cpu_baseline_ref"] = last_cpu_baseline_on_file()
        assert line["n_gpus"] == args.gpus == comm["world_size"]
        emit_line(line)


def last_cpu_baseline_on_file():
    """the newest `cpu_baseline` object of an N = 1 line kept in the repository: the driver's BENCH_rNN.json records, else
    profiles/rNN_bench.json (None when there is none)"""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")), reverse=True) + sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")), reverse=True)
    for path in cands:
        try:
            with open(path) as f:
                js = json.load(f)
        except (OSError, ValueError):
            continue
        js = js.get("parsed", js) if isinstance(js, dict) else {}
        cb = js.get("cpu_baseline") if isinstance(js, dict) else None
        if isinstance(cb, dict) and "value" in cb and js.get("n_gpus", 1) == 1:
            return dict(cb, source=os.path.relpath(path, ROOT))
    return None


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (reference tools/eval_glass.py:199-206
    `launch(main, num_gpus, ...)`), each a re-exec of this command with the torch.distributed.run environment, and return
    their exit code.  Refuses - non-zero - when the node cannot give every rank its own GPU (RCCL needs one device per rank);
    GLASS_BENCH_BACKEND=gloo lifts that check to exercise the N-rank plumbing on a 1-GPU box."""
    from glass_amd.distributed import launch_local_ranks, preflight_report
    backend = os.environ.get("GLASS_BENCH_BACKEND", "nccl")
    # first line of a multi-GPU run, before anything can hang: what this node offers (stderr; stdout is the ONE JSON line)
    print("[bench preflight] " + json.dumps(preflight_report(args.gpus, "gloo" if os.environ.get("GLASS_BENCH_DRYRUN") else backend)),
          file=sys.stderr, flush=True)
    if os.environ.get("GLASS_BENCH_DRYRUN"):                      # plumbing only: no GPU needed, gloo
        return launch_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no CPU fallback for the product path)", file=sys.stderr)
        return 2
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < args.gpus:
        print(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible; RCCL needs one device per rank - refusing to run fewer "
              f"ranks than asked (GLASS_BENCH_BACKEND=gloo exercises the {args.gpus}-rank path on fewer devices)", file=sys.stderr)
        return 2
    return launch_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)


_REAL_STDOUT = None


def emit_line(line: dict) -> None:
    """the ONE line of the contract, to the process's real stdout (see main: fd 1 is stderr while the run lasts)"""
    data = (json.dumps(line) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, data)
    else:
        os.write(_REAL_STDOUT, data)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))                       # this process only launches; the ranks print the line
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL prints a version banner when a
    # communicator is created, gloo its connection chatter), so file descriptor 1 is pointed at stderr for the whole run and the
    # line goes to the saved descriptor at the very end (emit_line)
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("GLASS_BENCH_STDOUT_NOISE"):                # (test hook: what a chatty native library does)
        os.write(1, b"noise from a native library on file descriptor 1\n")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    if os.environ.get("GLASS_BENCH_DRYRUN"):
        dist = None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)                                         # gloo's connection chatter goes to stderr
            try:
                from glass_amd.distributed import init_process_group
                if os.environ.get("GLASS_BENCH_DRYRUN_ABSENT_RANK") == str(rank):
                    time.sleep(600)                               # (test hook: a rank that never reaches the rendezvous)
                init_process_group("gloo")
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        from glass_amd.distributed import pin_to_gpu_numa_node
        print(f"[bench rank {rank}] LOCAL_RANK {local_rank} -> (dry run, no device) " + json.dumps(pin_to_gpu_numa_node(local_rank)),
              file=sys.stderr, flush=True)
        return dry_run(args, rank, world, dist)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("GLASS_BENCH_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" only to
    if backend == "nccl" and world > ndev:                        # exercise the N>1 plumbing on a 1-GPU box
        raise SystemExit(f"WORLD_SIZE={world} ranks but {ndev} GPU(s) visible: RCCL needs one device per rank")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # GLASS_BENCH_RCCL_WORLD1=1: a ONE-rank process group on the RCCL backend - communicator creation, the device-side barrier and
    # the per-step all_gather_into_tensor of the word records execute on this GPU (with one rank the collective is a copy).  The
    # only way to put RCCL itself through bench.py's code path on a 1-GPU lease; `n_gpus` stays 1.
    if world > 1 or os.environ.get("GLASS_BENCH_RCCL_WORLD1") == "1":
        import torch.distributed as dist
        from glass_amd.distributed import free_port, init_process_group, pin_to_gpu_numa_node
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # per rank, before the model is built: which device this rank drives and which CPUs it was pinned to (its GPU's NUMA node)
        print(f"[bench rank {rank}] LOCAL_RANK {local_rank} -> cuda:{dev_index} ({torch.cuda.get_device_name(dev_index)}) "
              + json.dumps(pin_to_gpu_numa_node(dev_index)), file=sys.stderr, flush=True)
        if backend == "nccl":
            init_process_group("nccl", device=dev)               # 120 s rendezvous / collective timeout: fail fast, not in 10 minutes
        else:
            # gloo's C++ side prints "[Gloo] Rank r is connected ..." to stdout: keep stdout for the ONE JSON line
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                init_process_group(backend)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

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
    # NSETS distinct input sets (images AND word boxes), cycled step by step: a step that kept state from its predecessor, or
    # that only ran fast on cache-warm inputs, would show.  Set s of global image g uses seed g + 1000 s.
    NSETS = DISTINCT_STEPS
    image_sets = [[make_image(g + 1000 * s, args.side, args.side).permute(2, 0, 1).float().contiguous().to(dev) for g in gidx]
                  for s in range(NSETS)]
    box_sets = [[make_boxes(g + 1000 * s, args.rois, args.side, args.side).to(dev) for g in gidx] for s in range(NSETS)]
    input_sets = [[{"image": im} for im in images] for images in image_sets]
    images, inputs = image_sets[0], input_sets[0]
    # GlassRunner's front-end (reference glass_runner.py:123-148): uint8 HWC image on the host -> device -> float CHW.
    # The synthetic float images above hold integer values 0..255, so the uint8 route gives bit-identical inputs.
    host_u8_sets = [[im.permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).cpu().contiguous().pin_memory() for im in ims]
                    for ims in image_sets]
    max_det = cfg.TEST.DETECTIONS_PER_IMAGE
    post = build_post_processor(cfg)                              # PostProcessorAcademic (device kernel)
    out_sizes = [(args.side, args.side)] * B
    steps_txt = cfg.MODEL.ROI_RECOGNIZER_HEAD.MAX_WORD_LENGTH + 1

    if args.workload == "backbone":
        il = model.preprocess_image(inputs)

    # GlassRunner's resize policy for this image size (reference glass_runner.py:111-121 on the cfg the bench loads:
    # MIN_SIZE_TEST 1200 / MAX_SIZE_TEST 2000 / MAX_UPSCALE_RATIO 3 -> a 1000 x 1000 image is UPSCALED x1.2)
    _m = float(args.side)
    policy_ratio = (cfg.INPUT.MAX_SIZE_TEST / _m if _m > cfg.INPUT.MAX_SIZE_TEST else
                    min(cfg.INPUT.MAX_UPSCALE_RATIO, cfg.INPUT.MIN_SIZE_TEST / _m) if _m < cfg.INPUT.MIN_SIZE_TEST else 1.0)
    policy_side = int(round(policy_ratio * args.side))
    policy_boxes = [[b * torch.tensor([policy_ratio, policy_ratio, policy_ratio, policy_ratio, 1.0], device=dev) for b in bs] for bs in box_sets]
    # the injected word boxes are INPUTS of the synthetic workload, resident in HBM like the images: their padded batch form is
    # built once per input set, not once per step (it was ~30 fill / copy launches inside every timed step)
    from glass_amd.modeling.fusion.recognizers_hybrid_head import prepare_injected_boxes
    if args.workload == "e2e":
        box_sets = [prepare_injected_boxes(bs, dev) for bs in box_sets]
        policy_boxes = [prepare_injected_boxes(bs, dev) for bs in policy_boxes]
    policy_scale = torch.tensor([[1.0 / policy_ratio, 1.0 / policy_ratio]] * B, dtype=torch.float32, device=dev)

    def local_step_g(from_host=False, s=0, keep=None, policy=False):
        """one step (input set `s`) as a generator (glass_amd/utils/pipeline.py): yields where the host reads counts back"""
        host_u8, boxes = host_u8_sets[s], box_sets[s]
        scale_xy = None
        if policy:
            # the runner's front end: H2D of the uint8 HWC images + ONE fused convert + bilinear resize kernel to the policy size;
            # the injected word boxes live in the resized frame, the results are un-scaled by 1 / ratio in the word post-processor
            step_inputs = [{"image": K.image_u8hwc_to_chw(h.to(dev, non_blocking=True), (policy_side, policy_side)),
                            "height": policy_side, "width": policy_side} for h in host_u8]
            boxes, scale_xy = policy_boxes[s], policy_scale
        elif from_host:
            # H2D of the uint8 HWC images (3 MB each, pinned -> stream-ordered) + fused convert on the step's stream.  NO resize:
            # this leg keeps the metric's 1000 x 1000 workload (the runner's policy WOULD upscale it - that is `runner_policy`)
            step_inputs = [{"image": K.image_u8hwc_to_chw(h.to(dev, non_blocking=True), (args.side, args.side))} for h in host_u8]
        else:
            step_inputs = input_sets[s]
        if args.workload == "backbone":                           # BASELINE configs[1]: trunk + FPN only
            model.backbone.forward_nhwc(il.nhwc4)                 # (the layers' weights carry the model's precision)
            return torch.zeros((B, 1), device=dev)
            yield                                                 # pragma: no cover (makes this a generator)
        out = yield from model.inference_g(step_inputs, override_boxes=boxes)   # list[{"instances": Instances}] (views)
        det = out.batch                                           # + this step's padded device-resident batch
        # word post-processing (merge, thresholds, polygons, text decode + text-score filter) for the 8 images
        words = yield from post.process_padded_g(det.boxes, det.scores, det.counts_dev, det.text, scale_xy, out_sizes,
                                                 {"orientations": det.orient})
        if keep is not None:                                      # the un-timed self-check: the step's character probabilities
            keep.append(det.text.clone())
        return pack_words(words.words, max_det, steps_txt)        # fixed-size per-image word records

    def step_g(from_host=False, s=0, keep=None, policy=args.runner_policy):
        rec = yield from local_step_g(from_host, s, keep, policy)
        if dist is not None and backend != "nccl":
            return all_gather_records(rec.cpu(), rows=B)
        return all_gather_records(rec, rows=B)                    # every rank holds B images: no count exchange needed

    def local_step(s=0):
        return drive(local_step_g(s=s, policy=args.runner_policy))

    def run_steps(n, from_host=args.from_host, depth=None, first=0, policy=args.runner_policy):
        """n steps over the input sets in turn, `--pipeline` of them in flight (each on its own stream; host segments
        interleaved in a fixed order, so every rank issues its all_gathers in the same order)"""
        return run_pipelined([(lambda s=(first + i) % NSETS: step_g(from_host, s, None, policy)) for i in range(n)],
                             depth=depth or args.pipeline, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # not part of W or K: the first step on each pipeline stream fills that stream's allocator pool (hipMalloc); one step
    # per stream so that a small --warmup cannot leave a cold one.  (Weights were packed in load_state_dict.)
    run_steps(args.pipeline)
    # once per run, outside the timed region: every input set's PIPELINED step returns bit for bit the record of its
    # synchronous run (per-step state / stream hazards would show here; tests/test_gpu_z_pipeline.py holds it for small shapes)
    pipelined_equals_sync = None
    if args.workload == "e2e":
        sync_txt = [[] for _ in range(NSETS)]
        sync_recs = [drive(step_g(args.from_host, s, sync_txt[s])) for s in range(NSETS)]
        pipe_txt = [[] for _ in range(2 * NSETS)]
        piped = run_pipelined([(lambda i=i: step_g(args.from_host, i % NSETS, pipe_txt[i])) for i in range(2 * NSETS)],
                              depth=args.pipeline, device=dev)
        torch.cuda.synchronize()
        # (with random weights the word records are mostly empty - every word fails the text-score threshold - so the
        #  check that has teeth is the one on the [R, 26, 97] character probabilities)
        pipelined_equals_sync = all(torch.equal(piped[i], sync_recs[i % NSETS]) and torch.equal(pipe_txt[i][0], sync_txt[i % NSETS][0])
                                    for i in range(2 * NSETS))
        if NSETS > 1:
            assert not torch.equal(sync_txt[0][0], sync_txt[1][0]), "the input sets must differ"
        assert pipelined_equals_sync, "a pipelined step's output differs from its synchronous run"
        del sync_txt, pipe_txt, piped
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
    timed_results = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(35)
    comm = {"world_size": 1, "backend": None}
    # the last timed step's gathered records: [world, B, record] on every rank, one count-prefixed record per image
    last = timed_results[-1]
    if args.workload == "e2e":
        assert last.dim() == 3 and last.shape[0] * last.shape[1] == world * B, (tuple(last.shape), world, B)
        assert bool((last[..., 0] >= 0).all()) and bool((last[..., 0] <= max_det).all()), "a gathered record carries a bad word count"
    if dist is not None:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        per_rank = torch.empty((world,), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(per_rank, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rccl = None
        if backend == "nccl":
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:                                  # noqa: BLE001 - a report field, never fatal
                rccl = f"unavailable ({type(e).__name__})"
        comm = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
                "per_rank_ms_per_step": [round(v / args.steps * 1e3, 3) for v in per_rank.tolist()],
                "gathered_records_shape": list(last.shape), "gathered_records_expected": world * B,
                "devices_visible": ndev, "rank_device": [r % ndev for r in range(world)]}
        assert comm["world_size"] == world
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    def timed_loop(n, **kw):
        """same protocol as the main loop (barrier + synchronize on both sides, max over ranks) for the side measurements"""
        run_steps(2, **kw)
        barrier()
        t0 = time.perf_counter()
        run_steps(n, **kw)
        barrier()
        d = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([d], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    extras = {}
    if not args.no_extras and args.workload == "e2e":
        n_x = max(4, min(args.steps, 20))
        if not args.from_host:
            d = timed_loop(n_x, from_host=True)
            extras["from_host"] = {"value": world * B * n_x / d, "unit": "images/sec", "ms_per_step": d / n_x * 1e3, "steps": n_x,
                                   "what": "same step starting from uint8 HWC images in pinned host memory: H2D copy (PCIe) + "
                                           "on-device u8 -> f32 CHW conversion inside the timed step (GlassRunner front-end)"}
        if not args.runner_policy and policy_ratio != 1.0:
            d = timed_loop(n_x, policy=True)
            extras["runner_policy"] = {"value": world * B * n_x / d, "unit": "images/sec", "ms_per_step": d / n_x * 1e3, "steps": n_x,
                                       "resize_ratio": policy_ratio, "model_input": f"{policy_side}x{policy_side} (padded to /32)",
                                       "what": "GlassRunner's own path at this image size: uint8 HWC host images -> H2D -> fused convert + "
                                               f"bilinear resize x{policy_ratio:g} (reference glass_runner.py:111-148 under this cfg's MIN_SIZE_TEST "
                                               f"{cfg.INPUT.MIN_SIZE_TEST}) -> the same step on the larger image -> results un-scaled; a bigger "
                                               "workload than the metric's (504 vs 423 GMAC per image), PCIe-inclusive"}
            d = timed_loop(n_x, policy=True, depth=1)
            extras["runner_policy"]["latency_ms_per_step"] = d / n_x * 1e3
        if args.precision == "fp32" and getattr(model.routing, "split", 0):
            # the same steps with the 1x1 layers back on the fp32-MFMA kernels (Routing.split = 0 on the model's own routing
            # object, which every layer's ConvWeight carries; both packed forms were built at load): the number to quote if
            # exact fp32 products formed on the bf16 matrix cores are not accepted as fp32 arithmetic
            split_was = model.routing.split
            model.routing.split = 0
            try:
                n_s = max(4, min(args.steps, 60))
                d = timed_loop(n_s)
                extras["fp32_mfma_only"] = {"value": world * B * n_s / d, "unit": "images/sec", "ms_per_step": d / n_s * 1e3, "steps": n_s,
                                            "what": "GLASS_PW_SPLIT=0: every convolution on v_mfma_f32_* (the 1x1 layers on conv1x1_pw_f32 / "
                                                    "conv_igemm_f32 instead of conv1x1_pw_split); same model, same inputs, same run"}
            finally:
                model.routing.split = split_was
        d = timed_loop(n_x, depth=1)
        extras["latency_ms_per_step"] = d / n_x * 1e3          # one step at a time: what a single request waits
        extras["latency_note"] = (f"{n_x} steps run one at a time (--pipeline 1); `value` keeps {args.pipeline} steps in flight, so its "
                                  f"ms_per_step is the throughput interval, not the latency")

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
        # the same three metered steps with the 1x1 layers on the fp32 MFMA: the split kernel sits at the package power cap and
        # lowers the clock the WHOLE step is granted, so the other families' per-launch times (and the dominant kernel's
        # `frac`) read ~3 % worse beside it than they do without it - both readings go on the line
        fam_fp32 = None
        if args.precision == "fp32" and getattr(model.routing, "split", 0):
            split_was = model.routing.split
            model.routing.split = 0
            try:
                with ConvMeter(K) as meter0:
                    for rep in range(3):
                        if rep:
                            meter0.new_step()
                        local_step()
                    fam_fp32 = meter0.summary()
            finally:
                model.routing.split = split_was
        model.roi_heads.two_stream_local = two
        conv_ms = sum(f["ms"] for f in fam.values())
        conv_flops = sum(f["algo_flops"] for f in fam.values())
        n_launch = sum(f["launches"] for f in fam.values())
        KNAME = {"winograd43": "conv3x3_wino43_f32 (Winograd F(4x4,3x3), fp32 MFMA 16x16x4, 16 tiles x 128 channels per workgroup)",
                 "winograd128": "conv3x3_wino128_f32 (Winograd F(2x2,3x3), fp32 MFMA, 32 tiles x 128 channels per workgroup)",
                 "winograd": "conv3x3_wino_f32 (Winograd F(2x2,3x3), fp32 MFMA, 64 tiles x 64 channels per workgroup)",
                 "pointwise": "conv1x1_pw_f32 (fp32 MFMA 16x16x4 weight-streaming 1x1 GEMM)",
                 "pointwise_split": "conv1x1_pw_split (1x1 GEMM, exact fp32 products as nine bf16 piece products: bf16 MFMA 16x16x32, fp32 accumulate; executed_* count the bf16 MFMA FLOP against the bf16 peak)",
                 "direct": "conv_igemm_f32 (fp32 MFMA implicit-GEMM conv/linear)",
                 "direct_fp16": "conv_igemm_f32<..., HALF> (fp16 MFMA implicit-GEMM conv/linear, fp32 accumulate)",
                 "packed_fp16": "conv_h16_kernel (fp16 MFMA 16x16x32 weight-streaming implicit-GEMM conv on fp16 tensors, fp32 accumulate)",
                 "fused_stem": "backbone_stem_fused_kernel + local_stem_fused_kernel (conv + ReLU + max-pool fused stems, fp32 MFMA 32x32x2)"}
        PKEY = {"winograd43": "conv3x3_wino43_f32", "winograd128": "conv3x3_wino128_f32", "winograd": "conv3x3_wino_f32", "pointwise": "conv1x1_pw_f32", "pointwise_split": "conv1x1_pw_split", "direct": "conv_igemm_f32_64x64",     # direct: its busiest instantiation
                "direct_fp16": "conv_igemm_f16", "packed_fp16": "conv_h16_kernel", "fused_stem": "backbone_stem_fused_kernel"}
        PEAK = PEAK_DEFAULT = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else FP16_MFMA_PEAK_TFLOPS
        dom = max(fam, key=lambda k: fam[k]["ms"])            # the dominant kernel by time
        # PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate
        # rocprofv3 --pmc runs, gfx950 x2 read correction applied; scripts/pmc_make_summary.py)
        pmc_all, pmc_file, pmc_note = {}, None, "no PMC summary under profiles/"
        from glass_amd._lib import source_sha16
        import glob
        # fp32: r*_pmc_conv_summary.json; the fp16-storage mode has its own passes (scripts/collect_profiles.sh step 3b)
        pmc_glob = "r*_pmc_conv_summary.json" if args.precision == "fp32" else f"r*_pmc_{args.precision}_summary.json"
        for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", pmc_glob)), reverse=True):
            with open(cand) as f:
                js = json.load(f)
            if js.get("lib_source_sha16") == source_sha16():
                pmc_all, pmc_file = js, os.path.relpath(cand, ROOT)
                pmc_note = f"HBM bytes/launch of this kernel, PMC ({pmc_file}; taken with this library: sha {source_sha16()})"
                break
            pmc_note = (f"null: {os.path.relpath(cand, ROOT)} was taken with library sha {js.get('lib_source_sha16')}, this run uses "
                        f"{source_sha16()} - re-run scripts/collect_profiles.sh")

        def fam_entry(k):
            f = fam[k]
            if f["launches"] == 0:
                return None
            sec = f["ms"] * 1e-3
            PEAK = FP16_MFMA_PEAK_TFLOPS if k == "pointwise_split" else PEAK_DEFAULT      # bf16 peak = fp16 peak
            pj = pmc_all.get(PKEY[k], {}) if isinstance(pmc_all.get(PKEY[k], {}), dict) else {}
            return {"kernel": KNAME[k], "launches_per_step": f["launches"], "kernel_ms_per_step": f["ms"],
                    "avg_launch_ms": f["ms"] / f["launches"],
                    "executed_tflops": f["exec_flops"] / sec / 1e12, "executed_frac": f["exec_flops"] / sec / 1e12 / PEAK,
                    "algorithmic_tflops": f["algo_flops"] / sec / 1e12,
                    "algorithmic_frac": f["algo_flops"] / sec / 1e12 / PEAK_DEFAULT,
                    "algorithmic_gflop_per_launch": f["algo_flops"] / f["launches"] / 1e9,
                    "traffic": pj.get("hbm_bytes_per_launch_corrected"), "mfma_util_percent_pmc": pj.get("MfmaUtil_percent"),
                    # the same family's weighted average launch in the committed rocprofv3 --kernel-trace --stats table
                    # (profiles/rNN_kernel_stats_serial.txt, folded into the PMC summary of this library): compare with
                    # avg_launch_ms * 1e3, which comes from HIP events inside this very run
                    "rocprof_avg_us": pj.get("rocprof_avg_us")}

        ent = fam_entry(dom)
        if fam_fp32 is not None and fam_fp32[dom]["launches"]:
            f0 = fam_fp32[dom]
            ent["with_fp32_mfma_only"] = {"kernel_ms_per_step": f0["ms"], "executed_frac": f0["exec_flops"] / (f0["ms"] * 1e-3) / 1e12 / PEAK_DEFAULT,
                                          "all_conv_ms_per_step": sum(f["ms"] for f in fam_fp32.values()),
                                          "what": "the same kernel metered in steps whose 1x1 layers run on the fp32 MFMA (GLASS_PW_SPLIT=0): the "
                                                  "split kernel's power draw costs every other kernel of the step clock"}
        other = [fam_entry(k) for k in fam if k != dom and fam[k]["launches"]]
        line = {
            "metric": "images/sec/GPU end-to-end spotting, 1000x1000, ~32 RoIs; 1/2/4/8 GPU scaling",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else
                     ("f16 operands, f32 accumulate/storage (reduced precision: not the headline)" if args.precision == "fp16" else
                      "f16 operands AND f16 activation storage on the conv path, f32 accumulate (reduced precision: not the headline)"),
            "data": "synthetic",
            "config": {"workload": ((("BASELINE.json configs[2]" if (args.side, args.rois) == (SIDE, ROIS) else
                                      "the pipeline of BASELINE.json configs[2] at another shape (configs[4]: 1333-long-side, 100 RoIs)")
                                     + ": backbone + RotatedROIAlign + recognition head, "
                                     f"{args.rois} RoIs/img, bs={B}/GPU, {args.side}x{args.side} (padded to /32), {args.precision}")
                                    if args.workload == "e2e" else
                                    f"BASELINE.json configs[1]: ResNet50-FPN backbone only, bs={B}/GPU, {args.side}x{args.side}, "
                                    f"{args.precision}"),
                       "images_per_gpu_per_step": B, "rois_per_image": args.rois, "proposals_per_image": 100,
                       "weights": "random-init (seed 1234), reference architecture",
                       "parallelism": f"image-shard x{world}, 1 all_gather of result records/step",
                       "steps_in_flight": args.pipeline, "distinct_steps": NSETS,
                       "pipelined_step_equals_synchronous_step": pipelined_equals_sync},
            "comm": comm,
            "images_per_sec_per_gpu": value / world,
            "hbm_peak_reserved_gb": torch.cuda.max_memory_reserved(dev) / 1e9,   # rank 0, whole run (of 288 GB)
            "roofline": {"bound": "mfma", "kernel": ent["kernel"],
                         # `achieved` = the FLOP this kernel really issues to the matrix cores / its measured time, `frac` =
                         # achieved / peak <= 1: the MFMA utilisation, to be compared with `mfma_util_percent_pmc` / 100
                         # (SQ_VALU_MFMA_BUSY_CYCLES of the same kernel family).  The Winograd kernels issue 4x (F(4x4,3x3))
                         # / 2.25x (F(2x2,3x3)) fewer multiplies than the direct-convolution count SURVEY 8d prices a layer
                         # at; that ALGORITHMIC rate is reported next to it (`algorithmic_tflops`, `algorithmic_frac` - may
                         # exceed 1 - and `algorithmic_speedup_vs_direct` = algorithmic / executed FLOP).
                         "achieved": ent["executed_tflops"], "peak": PEAK, "unit": "TFLOP/s",
                         "frac": ent["executed_frac"],
                         "peak_sustained_measured": FP32_MFMA_SUSTAINED_TFLOPS if args.precision == "fp32" else None,
                         "frac_of_sustained_peak": (ent["executed_tflops"] / FP32_MFMA_SUSTAINED_TFLOPS) if args.precision == "fp32" else None,
                         "achieved_is": "executed MFMA FLOP / kernel time (utilisation); the direct-convolution (algorithmic) rate is in algorithmic_*",
                         "algorithmic_tflops": ent["algorithmic_tflops"], "algorithmic_frac": ent["algorithmic_frac"],
                         "algorithmic_speedup_vs_direct": ent["algorithmic_tflops"] / ent["executed_tflops"],
                         "executed_tflops": ent["executed_tflops"], "executed_frac": ent["executed_frac"],
                         "traffic": ent["traffic"],
                         "traffic_note": pmc_note,
                         # the HBM side of the same kernel: PMC bytes per launch / its average launch time, against the 8 TB/s
                         # nominal peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming kernel sustains) - with `frac` this
                         # says which roofline, if any, binds: both well below 1 = issue / latency / power-bound
                         "hbm_tbps": (ent["traffic"] / (ent["avg_launch_ms"] * 1e-3) / 1e12) if ent["traffic"] else None,
                         "hbm_frac_of_peak": (ent["traffic"] / (ent["avg_launch_ms"] * 1e-3) / 8e12) if ent["traffic"] else None,
                         "mfma_util_percent_pmc": ent["mfma_util_percent_pmc"],
                         "launches_per_step": ent["launches_per_step"],
                         "algorithmic_gflop_per_launch": ent["algorithmic_gflop_per_launch"],
                         "avg_launch_ms": ent["avg_launch_ms"], "rocprof_avg_us": ent["rocprof_avg_us"],
                         "kernel_ms_per_step": ent["kernel_ms_per_step"],
                         "share_of_step": ent["kernel_ms_per_step"] / ms_per_step,
                         "frac_with_fp32_mfma_only": (ent.get("with_fp32_mfma_only") or {}).get("executed_frac"),
                         "with_fp32_mfma_only": ent.get("with_fp32_mfma_only"),
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
                                    "frac": (value / world) / (FP32_MFMA_PEAK_TFLOPS / (conv_flops / B / 1e12)),
                                    # the REFERENCE's direct-convolution count for this configuration (422.9 GMAC =
                                    # 0.846 TFLOP per image at 1024^2, P = 100, R = 32 -> 186 images/s/GPU): the step above
                                    # issues fewer convolution FLOP than that where it restructures exactly (P2P3 fusion on
                                    # the pooled bins instead of the whole map), so this is the fixed yardstick across rounds
                                    "survey_tflop_per_image": SURVEY_TFLOP_PER_IMAGE if (args.side, args.rois, args.workload) == (SIDE, ROIS, "e2e") else None,
                                    "frac_vs_survey": ((value / world) / (FP32_MFMA_PEAK_TFLOPS / SURVEY_TFLOP_PER_IMAGE))
                                    if (args.side, args.rois, args.workload) == (SIDE, ROIS, "e2e") else None},
        }
        line.update(extras)
        if args.precision == "fp32":
            line["arithmetic"] = ("fp32 operands, exact fp32 products, fp32 accumulation everywhere.  Convolutions on v_mfma_f32_16x16x4 / "
                                  "32x32x2 (Winograd, implicit GEMM, stems)" +
                                  ("; the 1x1 layers with Cin % 32 == 0, Cout % 128 == 0 form each fp32 product as the sum of its "
                                   "nine exact bf16-piece products (v = h + m + l, 8 + 8 + 8 significant bits; "
                                   "v_mfma_f32_16x16x32_bf16, fp32 accumulate): the same sum of exact products in another order - "
                                   "csrc/pointwise_split.hip, DESIGN.md section 3; `fp32_mfma_only` is the same run without it"
                                   if getattr(model.routing, "split", 0) == 9 else
                                   "; GLASS_PW_SPLIT=6: the 1x1 layers DROP the three bf16-piece products below 2^-23 of each product "
                                   "(opt-in, measurement only: not the exact product, not the headline)"
                                   if getattr(model.routing, "split", 0) == 6 else ""))
            # (the contract's `dtype: f32` line says where it is NOT the fp32 MFMA that multiplies)
            line["config"]["matrix_pipes"] = ("3x3 / stems / implicit-GEMM layers: fp32 MFMA; eligible 1x1 layers: " +
                                              {9: "bf16 MFMA on exact 3-way operand splits, all nine piece products (exact fp32 products)",
                                               6: "bf16 MFMA on 3-way operand splits, SIX piece products (opt-in: not exact)"}.get(
                                                  getattr(model.routing, "split", 0), "fp32 MFMA"))
        line["lib_source_sha16"] = source_sha16()
        # the persistent recurrent kernels' in-kernel waits are bounded: a hand-off that gave up raises a sticky status word
        # (bit 0 BiLSTM, bit 1 decoder) instead of hanging the GPU - 0 over the whole run or the line is not printed
        line["recurrent_handoff_status"] = K.recurrence_status()
        assert line["recurrent_handoff_status"] == 0, "a persistent recurrent kernel gave up a hand-off: outputs of that step are garbage"
        # weights are packed in load_state_dict (ops.native.ConvWeight); a launch that had to pack at launch time would
        # mean a layer / precision the loader did not foresee - 0 on this workload
        line["conv_launches_that_packed_weights_at_launch"] = K.packs_on_the_fly()
        if args.from_host:
            line["config"]["inputs"] = "uint8 HWC in pinned host memory (--from-host): PCIe-inclusive, NOT the contract's value"
        if args.runner_policy:
            line["config"]["inputs"] = (f"--runner-policy: uint8 HWC host images, resized x{policy_ratio:g} to {policy_side}x{policy_side} by the "
                                        "runner's policy: PCIe-inclusive and a LARGER workload than the metric's, NOT the contract's value")
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, args.cpu_side, args.rois)
        elif world > 1:
            # the CPU baseline is timed on rank 0 of the N = 1 run only (it needs the host cores to itself); an N > 1 line
            # carries the most recent N = 1 measurement on file as `cpu_baseline_ref`, named by its source
            line["cpu_baseline_ref"] = last_cpu_baseline_on_file()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        assert line["n_gpus"] == args.gpus == line["comm"]["world_size"], (line["n_gpus"], args.gpus, line["comm"])
        emit_line(line)


if __name__ == "__main__":
    main()

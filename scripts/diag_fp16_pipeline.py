"""Root-cause aid for VERDICT r2 #1: an fp16-mode step pipelined beside fp32-mode steps was not bit-identical to its
synchronous run on the driver box (tests/test_gpu_z_pipeline.py::test_conv_precision_does_not_leak_between_in_flight_steps).

  python scripts/diag_fp16_pipeline.py --trials 50                 failure rate of the test's own schedule
  python scripts/diag_fp16_pipeline.py --trials 50 --hook          every ops.native call's outputs are cloned in stream
                                                                   order; the FIRST diverging launch of a failing step is named
  --order fp16 | fp32 | mixed     which models are interleaved          --prio same   both pipeline streams in the normal class
  --h16 0                         fp16 mode without conv_h16_kernel     --depth N
Result (profiles/r03_fp16_pipeline_rootcause.log): a packed-f32 / double-rate-MFMA hardware interaction, docs/DESIGN_history_r1-r3.md section 4.
(Round 2 also had `--fill`: 600 dummy entries in the then-global packed-weight cache, to rule eviction out; the cache is gone.)
"""
import argparse
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=30)
ap.add_argument("--hook", action="store_true")
ap.add_argument("--order", default="mixed")
ap.add_argument("--prio", default="alt")
ap.add_argument("--h16", type=int, default=1)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--size", default="128x160")
ap.add_argument("--rois", type=int, default=4)
args = ap.parse_args()

import glass_amd  # noqa: E402
from glass_amd.config import get_glass_cfg  # noqa: E402
from glass_amd.ops import native as K  # noqa: E402
from glass_amd.utils import pipeline as P  # noqa: E402
from glass_amd.utils.synth import make_boxes, make_image, make_state_dict  # noqa: E402

dev = torch.device("cuda:0")
K.set_conv_h16(bool(args.h16))


def cfg(opts=()):
    return get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"] + list(opts))


sd = make_state_dict(1234)
models = {}
for prec in ("fp32", "fp16"):
    m = glass_amd.build_model(cfg(["MODEL.CONV_PRECISION", prec]))
    m.load_state_dict(sd)
    models[prec] = m
H, W = (int(v) for v in args.size.split("x"))
img = make_image(7, H, W).permute(2, 0, 1).float().contiguous().cuda()
boxes = [(make_boxes(7, args.rois, H, W) * torch.tensor([1, 1, 0.35, 0.5, 1.0])).cuda()]

LOG = None          # list the hooks append (name, [clones]) to; one per step


def install_hooks():
    skip = {"upload", "stream_handle", "set_conv_precision", "conv_precision", "act_dtype", "last_conv_path", "set_winograd",
            "set_winograd43", "set_conv_h16", "set_pointwise", "winograd_pack", "conv_out_size", "check", "lib", "pack_kblocked",
            "local_stem_supported", "linear"}
    for name, fn in list(vars(K).items()):
        if name.startswith("_") or name in skip or not callable(fn) or isinstance(fn, type) or getattr(fn, "__module__", "") != K.__name__:
            continue

        def wrap(fn=fn, name=name):
            def inner(*a, **kw):
                out = fn(*a, **kw)
                log = CUR.get("log")
                if log is not None:
                    ts = out if isinstance(out, (tuple, list)) else list(out.values()) if isinstance(out, dict) else [out]
                    extra = ""
                    if name == "conv2d_nhwc":
                        extra = f" {K.last_conv_path()} x{tuple(a[0].shape)} w{tuple(a[1].shape)}"
                    log.append((name + extra, [t.clone() for t in ts if isinstance(t, torch.Tensor)]))
                return out
            return inner
        setattr(K, name, wrap())


CUR = {}


def make(prec, logs=None, idx=None):
    def gen():
        body = models[prec].inference_g([{"image": img}], do_postprocess=False, override_boxes=boxes)
        if logs is None:
            out = yield from body
            return out.batch.text.clone()
        # the hooks must know which step's segment is executing: set it around every segment
        val, first = None, True
        while True:
            CUR["log"] = logs[idx]
            try:
                req = next(body) if first else body.send(val)
            except StopIteration as e:
                CUR["log"] = None
                return e.value.batch.text.clone()
            CUR["log"] = None
            first = False
            val = yield req
    return gen


order = {"mixed": ["fp16", "fp32", "fp16", "fp16", "fp32", "fp32", "fp16"], "fp16": ["fp16"] * 7, "fp32": ["fp32"] * 7}[args.order]
if args.prio == "same":
    P._STREAMS[(str(dev), args.depth)] = [torch.cuda.Stream(device=dev) for _ in range(args.depth)]

if args.hook:
    install_hooks()
ref, ref_log = {}, {}
for p in ("fp32", "fp16"):
    logs = [[]] if args.hook else None
    ref[p] = P.drive(make(p, logs, 0)())
    ref_log[p] = logs[0] if args.hook else None
    # synchronous determinism first: the same step again, alone
    again = P.drive(make(p)())
    print(f"sync {p}: repeat bit-identical: {torch.equal(again, ref[p])}", flush=True)
torch.cuda.synchronize()

fails = 0
per_idx = [0] * len(order)
for t in range(args.trials):
    logs = [[] for _ in order] if args.hook else None
    got = P.run_pipelined([make(p, logs, i) for i, p in enumerate(order)], depth=args.depth, device=dev)
    torch.cuda.synchronize()
    bad = [i for i, (p, g) in enumerate(zip(order, got)) if not torch.equal(g, ref[p])]
    if bad:
        fails += 1
        for i in bad:
            per_idx[i] += 1
        i = bad[0]
        d = (got[i] - ref[order[i]]).abs()
        print(f"trial {t}: steps {bad} differ; step {i} ({order[i]}): {int((d > 0).sum())} of {d.numel()} elements, max |d| {float(d.max()):.3e}", flush=True)
        if args.hook:
            rl, gl = ref_log[order[i]], logs[i]
            print(f"   launches: ref {len(rl)} got {len(gl)}")
            shown = 0
            for k, ((rn, rt), (gn, gt)) in enumerate(zip(rl, gl)):
                if rn != gn:
                    print(f"   op {k}: NAME differs {rn} vs {gn}")
                    break
                for j, (a, b) in enumerate(zip(rt, gt)):
                    if a.shape != b.shape or not torch.equal(a, b):
                        if a.shape == b.shape:
                            dd = (a.float() - b.float()).abs()
                            nz = (dd > 0).nonzero()
                            print(f"   op {k} {rn} out{j} {tuple(a.shape)} {a.dtype}: {int((dd > 0).sum())} of {dd.numel()} differ, max {float(dd.max()):.3e}; "
                                  f"first idx {nz[0].tolist()} last idx {nz[-1].tolist()}")
                            if int((dd > 0).sum()) <= 256:
                                fl = (dd.flatten() > 0).nonzero().flatten()
                                lo = fl[0].item() // 4 * 4
                                print("      flat idx of diffs:", fl.tolist()[:64])
                                print("      ref", [round(v, 4) for v in a.flatten()[lo:lo + 16].tolist()])
                                print("      got", [round(v, 4) for v in b.flatten()[lo:lo + 16].tolist()])
                        else:
                            print(f"   op {k} {rn} out{j}: shape {tuple(a.shape)} vs {tuple(b.shape)}")
                        shown += 1
                if shown >= 4:
                    break
print(f"RESULT order={args.order} prio={args.prio} h16={args.h16} hook={args.hook} depth={args.depth}: {fails} of {args.trials} trials failed; per step {per_idx}")

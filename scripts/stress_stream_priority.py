"""Is a kernel's result stable while ANOTHER stream of a different priority class keeps the chip busy?

Found while root-causing VERDICT r2 #1 (scripts/diag_fp16_pipeline.py): with two steps in flight on a high-priority and a
normal-priority HIP stream, single wavefronts of `roi_align_rotated_kernel` came back with lanes 48..63 of two result
registers wrong - no scratch, no LDS, no atomics in that kernel.  This script takes the model out of the picture:

  victim  (stream V): one deterministic kernel launched `--iters` times, every output compared with its solo result
                      kinds: roi (our RoIAlign), torch (pure-torch elementwise chain: no code of this repo at all)
  hammer  (stream H): back-to-back 3x3 convolutions (our F(4x4) kernel) or torch matmuls

  python scripts/stress_stream_priority.py --victim-prio 0 --hammer-prio -1     mixed classes
  python scripts/stress_stream_priority.py --victim-prio 0 --hammer-prio 0      same class
"""
import argparse
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--victim", default="roi")
ap.add_argument("--hammer", default="conv")
ap.add_argument("--victim-prio", type=int, default=0)
ap.add_argument("--hammer-prio", type=int, default=-1)
ap.add_argument("--iters", type=int, default=400)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()

from glass_amd.ops import native as K  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
sv = torch.cuda.Stream(device=dev, priority=args.victim_prio)
sh = torch.cuda.Stream(device=dev, priority=args.hammer_prio)

# victim inputs
img = torch.randn((1, 128, 160, 4), device=dev) * 50
img[..., 3] = 0
boxes = torch.tensor([[60., 50., 40., 20., 30.], [100., 80., 50., 16., -20.], [40., 90., 30., 12., 75.], [80., 30., 60., 25., 5.]], device=dev)
bidx = torch.zeros((4,), dtype=torch.int32, device=dev)
xt = torch.randn((1 << 20,), device=dev)


def victim():
    if args.victim == "roi":
        return K.roi_align_rotated([img], [1.0], boxes, bidx, (128, 128), 2, channels=4)
    return torch.sin(xt) * 2.0 + xt * xt


# hammer inputs
hx = torch.randn((8, 64, 64, 256), device=dev)
hw = torch.randn((256, 3, 3, 256), device=dev) * 0.05
ha = torch.randn((4096, 4096), device=dev)


hx16 = torch.randn((1, 32, 40, 256), device=dev)                 # the small maps of the failing test (128 x 160 image)
hw1 = torch.randn((256, 1, 1, 256), device=dev) * 0.05
hx4 = torch.randn((4, 128, 128, 4), device=dev)
hw4 = torch.randn((16, 3, 3, 4), device=dev) * 0.05


def hammer():
    if args.hammer == "conv":
        K.conv2d_nhwc(hx, hw, None, padding=1)
    elif args.hammer == "mm":
        torch.mm(ha, ha)
    elif args.hammer in ("h16", "h16small", "h16pw", "f16t", "cast"):
        prev = K.set_conv_precision("fp16")
        try:
            if args.hammer == "h16":
                K.conv2d_nhwc(hx, hw, None, padding=1)           # cast + conv_h16_kernel<16,1>
            elif args.hammer == "h16small":
                K.conv2d_nhwc(hx16, hw, None, padding=1)         # cast + a small-grid variant
            elif args.hammer == "h16pw":
                K.conv2d_nhwc(hx16, hw1, None)                   # 1x1
            elif args.hammer == "f16t":
                K.conv2d_nhwc(hx4, hw4, None, padding=1)         # the fp32 template with fp16 operands (Cin = 4)
            else:
                xh = torch.empty(hx.shape, dtype=torch.float16, device=dev)
                import ctypes
                from glass_amd._lib import lib
                lib().glass_cast_f32_to_f16(ctypes.c_void_p(hx.data_ptr()), ctypes.c_void_p(xh.data_ptr()), ctypes.c_int64(hx.numel()),
                                            ctypes.c_void_p(K.stream_handle()))
        finally:
            K.set_conv_precision(prev)


ref = victim().clone()
hammer()
torch.cuda.synchronize()
total_bad = 0
for r in range(args.rounds):
    outs = []
    sv.wait_stream(torch.cuda.current_stream())
    sh.wait_stream(torch.cuda.current_stream())
    for i in range(args.iters):
        if args.hammer != "none":
            with torch.cuda.stream(sh):
                hammer()
        with torch.cuda.stream(sv):
            outs.append(victim())
    torch.cuda.synchronize()
    bad = 0
    for i, o in enumerate(outs):
        if not torch.equal(o, ref):
            bad += 1
            if bad <= 3:
                d = (o - ref).abs().flatten()
                nz = (d > 0).nonzero().flatten()
                print(f"  round {r} launch {i}: {nz.numel()} elements differ, flat idx {nz[0].item()}..{nz[-1].item()}, max {float(d.max()):.3e}", flush=True)
                a, b = nz[0].item() // 4 * 4, min(nz[0].item() // 4 * 4 + 16, d.numel())
                print("     ref", [round(v, 4) for v in ref.flatten()[a:b].tolist()])
                print("     got", [round(v, 4) for v in o.flatten()[a:b].tolist()], flush=True)
    total_bad += bad
    print(f"round {r}: {bad} of {args.iters} victim launches differ", flush=True)
print(f"RESULT victim={args.victim}@prio{args.victim_prio} hammer={args.hammer}@prio{args.hammer_prio}: {total_bad} of {args.rounds * args.iters} launches differ")

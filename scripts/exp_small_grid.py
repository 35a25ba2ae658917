"""One-image shapes: every Winograd form of a 3x3 layer + split-K counts of a direct layer, timed back to back (HIP events).
    python scripts/exp_small_grid.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


R0 = K.default_routing()
print("== 3x3 layers: us per launch by form")
for (N, H, W, Cin, Cout) in [(32, 16, 33, 256, 256), (32, 16, 33, 128, 256), (32, 8, 32, 512, 256), (1, 64, 64, 256, 256), (1, 32, 32, 512, 512),
                             (1, 128, 128, 128, 128), (1, 128, 128, 256, 256), (1, 256, 256, 64, 64), (32, 32, 32, 128, 128), (32, 4, 32, 256, 256)]:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = K.prepare_conv_weights(torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05, "fp32", ragged=True)
    y = torch.empty((N, H, W, Cout), device=dev)
    res = {}
    forms = [("auto", dict()), ("f43", dict(winograd="f43")), ("f43 full", dict(winograd="f43", routing=R0.replace(ragged=False))),
             ("f22 full", dict(winograd=True)), ("direct", dict(winograd=False, routing=R0.replace(splitk=False))), ("direct+splitK", dict(winograd=False))]
    if W % 2 == 1:
        forms.append(("f22 body+strip", dict(winograd="f22r")))
    for name, kw in forms:
        try:
            t = timeit(lambda: K.conv2d_nhwc(x, w, None, padding=1, out=y, **kw))
            res[name] = (t, K.last_conv_path())
        except Exception as e:   # noqa: BLE001
            res[name] = (float("nan"), type(e).__name__)
    print(f"[{N},{H},{W},{Cin}]->{Cout}: " + "  ".join(f"{k} {v[0]:.0f} ({v[1]})" for k, v in res.items()))
print("== direct layers: us per launch by split count (0 = single slice)")
import ctypes
for (N, H, W, Cin, Cout, k, st, pad) in [(100, 1, 1, 12544, 2048, 1, 1, 0), (100, 1, 1, 2048, 2048, 1, 1, 0), (100, 1, 1, 2048, 11, 1, 1, 0), (1, 64, 64, 256, 256, 3, 1, 1),
                                          (1, 32, 32, 512, 512, 3, 1, 1), (1, 32, 32, 2048, 512, 1, 1, 0), (1, 64, 64, 1024, 256, 1, 1, 0), (512, 1, 1, 1536, 256, 1, 1, 0),
                                          (1, 64, 64, 256, 1024, 1, 1, 0), (800, 1, 1, 12544, 2048, 1, 1, 0), (800, 1, 1, 2048, 11, 1, 1, 0)]:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = torch.randn((Cout, k, k, Cin), device=dev) * 0.05
    orig = K._splitk_slices
    out = []
    for s in (0, 2, 4, 8, 12, 16, 24, 28, 32):
        K._splitk_slices = lambda *a, s=s: s
        try:
            t = timeit(lambda: K.conv2d_nhwc(x, w, None, stride=st, padding=pad, winograd=False, routing=R0.replace(pw=False)), n=100)
            out.append(f"{s}: {t:.0f}")
        except Exception:   # noqa: BLE001
            pass
    K._splitk_slices = orig
    M = N * ((H + 2 * pad - k) // st + 1) * ((W + 2 * pad - k) // st + 1)
    print(f"[{N},{H},{W},{Cin}]->{Cout} k{k} s{st}: rule picks {orig(R0, M, k * k * Cin, Cin, Cout)} |  " + "  ".join(out))

"""box head fc1 ([800, 12544] x [12544, 2048]) and the other implicit-GEMM 1x1 leftovers: the routed kernel vs the forced bf16-split kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, H, W, Cin, Cout in (("fc1 800 rows", 800, 1, 1, 12544, 2048), ("fc2 800 rows", 800, 1, 1, 2048, 2048), ("fpn lat5 2048->256 @ 8x32x32", 8, 32, 32, 2048, 256),
                                 ("lstm proj 8192 x 512->256", 8192, 1, 1, 512, 256), ("fc1 100 rows (B=1)", 100, 1, 1, 12544, 2048)):
    x = torch.randn((N, H, W, Cin), device=dev)
    w = K.prepare_conv_weights(torch.randn((Cout, 1, 1, Cin), device=dev) * 0.02, "all")
    b = torch.randn((Cout,), device=dev)
    y0 = K.conv2d_nhwc(x, w, b, relu=1); p0 = K.last_conv_path()
    t0 = timeit(lambda: K.conv2d_nhwc(x, w, b, relu=1))
    try:
        y1 = K.conv2d_nhwc(x, w, b, relu=1, winograd="pws9"); t1 = timeit(lambda: K.conv2d_nhwc(x, w, b, relu=1, winograd="pws9"))
        err = float((y1 - y0).abs().max() / y0.abs().max())
    except Exception as e:
        t1, err = float("nan"), str(e)[:60]
    fl = 2.0 * N * H * W * Cin * Cout
    print(f"{name:32s} routed ({p0}) {t0:.3f} ms {fl / t0 / 1e9:6.1f} TF/s | forced split {t1:.3f} ms {fl / t1 / 1e9:6.1f} TF/s | rel diff {err}")

import os, sys
sys.path.insert(0, "/root/repo/glass-text-spotting_amd")
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for res in (False, True):
    N, H, W, Cin, Cout = 32, 16, 33, 256, 256
    x = torch.randn((N, H, W, Cin), device=dev)
    w = K.prepare_conv_weights(torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05, "all", ragged=True)
    b = torch.randn((Cout,), device=dev)
    r = torch.randn((N, H, W, Cout), device=dev) if res else None
    f = lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1, residual=r, res_mode=1 if res else 0)
    K._TLS.force_f43k = 0
    y0 = f(); print("routed", K.last_conv_path(), round(timeit(f), 1))
    for fs in ((2, True), (4, True), (2, False)):
        K._TLS.force_f43k = fs
        y1 = f(); print(fs, K.last_conv_path(), round(timeit(f), 1), float((y1 - y0).abs().max() / y0.abs().max()))
    K._TLS.force_f43k = None
    for wino in ("f43", "f22r", True):
        try:
            f2 = lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1, residual=r, res_mode=1 if res else 0, winograd=wino)
            f2(); print("forced", wino, K.last_conv_path(), round(timeit(f2), 1))
        except Exception as e: print(wino, str(e)[:60])

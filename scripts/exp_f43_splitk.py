"""F(4x4) split-K (glass_conv3x3_winograd43_splitk_nhwc) on the 3x3 layers one image in flight leaves on a fraction of the chip: the routed kernel
vs every slice count, to fit ops.native._f43_splitk_plan.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
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
SHAPES = [("res4 3x3 256@64x64", 1, 64, 64, 256, 256, False), ("res5 3x3 512@32x32", 1, 32, 32, 512, 512, False), ("fusion 512->256@32x8x32", 32, 8, 32, 512, 256, False),
          ("FPN/RPN p3 256@128x128", 1, 128, 128, 256, 256, False), ("local l3 256@32x16x33 +res", 32, 16, 33, 256, 256, True), ("local l4 256@32x4x32 +res", 32, 4, 32, 256, 256, True),
          ("res3 128@128x128", 1, 128, 128, 128, 128, False), ("FPN p2 256@256x256", 1, 256, 256, 256, 256, False), ("res4 B2", 2, 64, 64, 256, 256, False)]
for name, N, H, W, Cin, Cout, res in SHAPES:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = K.prepare_conv_weights(torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05, "all", ragged=(W % 4 == 1))
    b = torch.randn((Cout,), device=dev)
    r = torch.randn((N, H, W, Cout), device=dev) if res else None
    f = lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1, residual=r, res_mode=1 if res else 0)
    K._TLS.force_f43k = 0
    y0 = f(); p0 = K.last_conv_path(); t0 = timeit(f)
    row = [f"{name:30s} routed-without ({p0}) {t0:6.1f} us |"]
    for sl in (2, 4, 8, 16):
        if (Cin // 32) % sl:
            continue
        K._TLS.force_f43k = sl
        try:
            y1 = f()
            assert K.last_conv_path() == "winograd43k", K.last_conv_path()
            err = float((y1 - y0).abs().max() / y0.abs().max())
            row.append(f" s{sl}: {timeit(f):6.1f} us ({err:.1e})")
        except Exception as e:
            row.append(f" s{sl}: {str(e)[:40]}")
    K._TLS.force_f43k = None
    f(); row.append(f" | model picks {K.last_conv_path()} {timeit(f):6.1f} us")
    print("".join(row), flush=True)

"""Micro-benchmark of glass_conv2d_nhwc on representative GLASS layers (GPU box only)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K

dev = torch.device("cuda:0")
LAYERS = [
    # name, N, H, W, Cin, Cout, k, stride, pad
    ("local l3 3x3 256 @16x33 R=256", 256, 16, 33, 256, 256, 3, 1, 1),
    ("fpn_out2 3x3 256 @256", 8, 256, 256, 256, 256, 3, 1, 1),
    ("rpn 3x3 256 @64", 8, 64, 64, 256, 256, 3, 1, 1),
    ("res2.conv3 1x1 64->256", 8, 256, 256, 64, 256, 1, 1, 0),
    ("res3.conv3 1x1 128->512", 8, 128, 128, 128, 512, 1, 1, 0),
    ("res4.conv3 1x1 256->1024", 8, 64, 64, 256, 1024, 1, 1, 0),
    ("fpn_lat2 1x1 256->256 @256", 8, 256, 256, 256, 256, 1, 1, 0),
    ("res2.conv2 3x3 64", 8, 256, 256, 64, 64, 3, 1, 1),
    ("res2.conv1 1x1 256->64", 8, 256, 256, 256, 64, 1, 1, 0),
    ("res3.conv2 3x3 128", 8, 128, 128, 128, 128, 3, 1, 1),
    ("res5.conv2 3x3 512", 8, 32, 32, 512, 512, 3, 1, 1),
    ("stem7x7", 8, 1024, 1024, 4, 64, 7, 2, 3),
    ("local conv0_1 3x3 4->16 @128", 256, 128, 128, 4, 16, 3, 1, 1),
    ("local conv0_2 3x3 16->32 @128", 256, 128, 128, 16, 32, 3, 1, 1),
    ("local l1 3x3 64 @64 R=256", 256, 64, 64, 64, 64, 3, 1, 1),
    ("local l2 3x3 128 @32 R=256", 256, 32, 32, 128, 128, 3, 1, 1),
    ("fusion out 3x3 512->256 R=256", 256, 8, 32, 512, 256, 3, 1, 1),
    ("fc1 800x12544->2048", 800, 1, 1, 12544, 2048, 1, 1, 0),
    ("rpn p6 3x3 256 @16", 8, 16, 16, 256, 256, 3, 1, 1),
    ("res3.conv1 1x1 512->128", 8, 128, 128, 512, 128, 1, 1, 0),
    ("res4.conv1 1x1 1024->256", 8, 64, 64, 1024, 256, 1, 1, 0),
    ("res5.conv3 1x1 512->2048", 8, 32, 32, 512, 2048, 1, 1, 0),
    ("fpn p5 3x3 256 @32", 8, 32, 32, 256, 256, 3, 1, 1),
    ("res5-like 3x3 512 @16 B=2", 2, 32, 32, 512, 512, 3, 1, 1),
    ("local tail 3x3 256 @4x32 R=64", 64, 4, 32, 256, 256, 3, 1, 1),
    ("rpn heads 1x1 256->72 @256", 8, 256, 256, 256, 72, 1, 1, 0),
    ("rpn heads 1x1 256->72 @128", 8, 128, 128, 256, 72, 1, 1, 0),
]
def timed(fn, it=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, N, H, W, Cin, Cout, k, s, p in LAYERS:
    if only and only not in name:
        continue
    x = torch.randn((N, H, W, Cin), device=dev)
    w = K.prepare_conv_weights(torch.randn((Cout, k, k, Cin), device=dev) * 0.05, "all")     # packed once, like a loaded model's layer
    b = torch.randn((Cout,), device=dev)
    K.set_pointwise(False)
    y = K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, winograd=False)
    ms = timed(lambda: K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, out=y, winograd=False))
    K.set_pointwise("all")
    flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k
    line = f"{name:36s} direct {ms:8.3f} ms {flops / ms / 1e9:7.1f} TFLOP/s"
    d = K.ConvDesc(N, H, W, Cin, Cout, k, k, s, s, p, p, y.shape[1], y.shape[2], Cin, Cout, 0, 1, 1, 0, 0)
    if k == 1 and K.lib().glass_pointwise_supported(K.ctypes.byref(d)):
        yp = torch.empty_like(y)
        msp = timed(lambda: K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, out=yp))
        err = float((yp - y).abs().max() / y.abs().max())
        line += f" | pointwise {msp:8.3f} ms {flops / msp / 1e9:7.1f} TFLOP/s  x{ms / msp:.2f}  rel.err {err:.1e}"
    if k == 3 and K.lib().glass_winograd_supported(K.ctypes.byref(d)):
        y2 = torch.empty_like(y)
        msw = timed(lambda: K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, out=y2, winograd=True))
        err = float((y2 - y).abs().max() / y.abs().max())
        line += f" | winograd {msw:8.3f} ms {flops / msw / 1e9:7.1f} TFLOP/s-equivalent  x{ms / msw:.2f}  rel.err {err:.1e}"
    if k == 3 and K.lib().glass_winograd43_supported(K.ctypes.byref(d)):
        y4 = torch.empty_like(y)
        ms4 = timed(lambda: K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, out=y4, winograd="f43"))
        err = float((y4 - y).abs().max() / y.abs().max())
        line += f" | F(4x4) {ms4:8.3f} ms {flops / ms4 / 1e9:7.1f} TF/s-eq  x{ms / ms4:.2f}  rel.err {err:.1e}"
    if os.environ.get("BENCH_CONV_NO_FP16"):
        print(line, flush=True)
        continue
    K.set_conv_precision("fp16")
    y3 = torch.empty_like(y)
    msh = timed(lambda: K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, out=y3))
    K.set_conv_precision("fp32")
    line += f" | fp16-mfma {msh:8.3f} ms x{ms / msh:.2f} vs fp32 direct"
    print(line, flush=True)

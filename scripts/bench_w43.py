"""F(4x4,3x3) kernel timing on the layers it takes (GPU box only): GLASS_W43_ABL=0..3 for the timing ablations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
LAYERS = [("fpn_out2 256@256 B8", 8, 256, 256, 256, 256), ("local l3 256@16x33 R256", 256, 16, 33, 256, 256),
          ("fusion 512->256@8x32 R256", 256, 8, 32, 512, 256), ("local l2 128@32 R256", 256, 32, 32, 128, 128),
          ("local l1 64@64 R256", 256, 64, 64, 64, 64), ("res2.conv2 64@256 B8", 8, 256, 256, 64, 64),
          ("local l1.0 32->64@64 R256", 256, 64, 64, 32, 64)]
if os.environ.get("W43_LAYERS"):            # e.g. W43_LAYERS=0,5: only those rows (PMC passes of one shape)
    LAYERS = [LAYERS[int(i)] for i in os.environ["W43_LAYERS"].split(",")]
for name, N, H, W, Cin, Cout in LAYERS:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = K.prepare_conv_weights(torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05, "all")     # packed once, like a loaded model's layer
    b = torch.randn((Cout,), device=dev)
    y = torch.empty((N, H, W, Cout), device=dev)
    f = lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1, out=y, winograd="f43")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ex = 2.0 * N * ((H + 3) // 4) * ((W + 3) // 4) * 36 * Cout * Cin
    print(f"ABL={os.environ.get('GLASS_W43_ABL', '0')} {name:28s} {ms:7.3f} ms  executed {ex / ms / 1e9:6.1f} TF/s ({ex / ms / 1e9 / 157.3:.3f} of peak)", flush=True)
if os.environ.get("W43_RES"):
    for name, N, H, W, Cin, Cout in LAYERS[:2]:
        x = torch.randn((N, H, W, Cin), device=dev); w = K.prepare_conv_weights(torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05, "all")
        b = torch.randn((Cout,), device=dev); y = torch.empty((N, H, W, Cout), device=dev); r = torch.randn((N, H, W, Cout), device=dev)
        f = lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1, out=y, residual=r, res_mode=1, winograd="f43")
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"ABL=res {name:28s} +residual {e0.elapsed_time(e1) / 10:7.3f} ms", flush=True)

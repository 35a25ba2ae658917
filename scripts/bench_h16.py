"""conv_h16_kernel timing on the layers of the fp16-storage mode (GPU box only).  With a library built with
-DGLASS_H16_ABLATIONS (GLASS_HIP_LIB=...), GLASS_H16_ABL=1..5 selects the timing ablations of csrc/conv_h16.hip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
# name, N, H, W, Cin, Cout, k, residual
LAYERS = [("fpn_out2 256@256 B8", 8, 256, 256, 256, 256, 3, 0), ("local l3 256@16x33 R256", 256, 16, 33, 256, 256, 3, 1),
          ("fpn_out3 256@128 B8", 8, 128, 128, 256, 256, 3, 0), ("local l2 128@32 R256", 256, 32, 32, 128, 128, 3, 1),
          ("local l1 64@64 R256", 256, 64, 64, 64, 64, 3, 1), ("res2.conv2 64@256 B8", 8, 256, 256, 64, 64, 3, 0),
          ("res2.conv3 64->256@256", 8, 256, 256, 64, 256, 1, 1), ("res2.conv1 256->64@256", 8, 256, 256, 256, 64, 1, 0),
          ("lateral2 256->256@256", 8, 256, 256, 256, 256, 1, 1), ("res4.conv1 1024->256@64", 8, 64, 64, 1024, 256, 1, 0),
          ("res5.conv2 512@32", 8, 32, 32, 512, 512, 3, 0)]
K.set_conv_precision("fp16s")
only = os.environ.get("H16_ONLY")
for name, N, H, W, Cin, Cout, k, res in LAYERS:
    if only and only not in name:
        continue
    x = torch.randn((N, H, W, Cin), device=dev).half()
    w = K.prepare_conv_weights(torch.randn((Cout, k, k, Cin), device=dev) * 0.05, "all")     # packed once, like a loaded model's layer
    b = torch.randn((Cout,), device=dev)
    y = torch.empty((N, H, W, Cout), device=dev, dtype=torch.float16)
    r = torch.randn((N, H, W, Cout), device=dev).half() if res else None
    out = []
    for packed in (True, False):
        K.set_conv_h16(packed)
        f = lambda: K.conv2d_nhwc(x, w, b, padding=k // 2, relu=1, out=y, residual=r, res_mode=1 if res else 0)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
        if os.environ.get("GLASS_H16_ABL"):
            break
    fl = 2.0 * N * H * W * k * k * Cout * Cin
    print(f"ABL={os.environ.get('GLASS_H16_ABL', '0')} {name:26s} packed {out[0]:7.3f} ms {fl / out[0] / 1e9:7.1f} TF/s"
          + (f"   fp32-template {out[1]:7.3f} ms {fl / out[1] / 1e9:7.1f} TF/s" if len(out) > 1 else ""), flush=True)

"""F(4x4,3x3) kernel against float64 on random layers: max |error| / output range (the figure winograd43.hip's header quotes and
tests/test_gpu_f_ops.py bounds at 2e-5).  Run once per library (GLASS_HIP_LIB) to compare transform point sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
for name, N, H, W, Cin, Cout, mean in (("256->256 random", 2, 64, 64, 256, 256, 0.0), ("256->256 post-ReLU inputs with a mean", 2, 64, 64, 256, 256, 1.0),
                                       ("512->256", 2, 32, 32, 512, 256, 0.0), ("64->64 (narrow)", 2, 64, 64, 64, 64, 0.0), ("128->128 wide-range x30", 2, 48, 48, 128, 128, 0.0)):
    g = torch.Generator().manual_seed(7)
    x = torch.randn((N, H, W, Cin), generator=g)
    if mean:
        x = torch.relu(x + mean)
    if "x30" in name:
        x = x * 30.0
    w = torch.randn((Cout, 3, 3, Cin), generator=g) * 0.05
    b = torch.randn((Cout,), generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    cw = K.prepare_conv_weights(w.to(dev), "all")
    y = K.conv2d_nhwc(x.to(dev), cw, b.to(dev), padding=1, winograd="f43").cpu().double()
    yd = K.conv2d_nhwc(x.to(dev), cw, b.to(dev), padding=1, winograd=False).cpu().double()
    rng = float(ref.abs().max())
    print(f"{os.path.basename(os.environ.get('GLASS_HIP_LIB', 'libglass_hip.so')):28s} {name:40s} F(4x4) max|err|/range {float((y - ref).abs().max()) / rng:.2e}   direct kernel {float((yd - ref).abs().max()) / rng:.2e}")

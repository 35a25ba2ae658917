"""Is the F(4x4) epilogue (8.7 us of a 110 us block at full grid) bound by the chip-wide store burst or by the CU's own store path?
GLASS_W43_ABL=4 (phase stamps, compiled in) on the same layer shape at grids of 32 ... 4096 workgroups (GPU box only)."""
import os, sys
os.environ["GLASS_W43_ABL"] = "4"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
w = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "all")
b = torch.randn((256,), device=dev)
for N, H, W, res in ((1, 64, 64, 0), (1, 128, 128, 0), (2, 128, 128, 0), (4, 256, 256, 0), (1, 64, 64, 1), (2, 128, 128, 1)):
    x = torch.randn((N, H, W, 256), device=dev)
    r = torch.randn((N, H, W, 256), device=dev) if res else None
    y = torch.empty((N, H, W, 256), device=dev)
    print(f"== N={N} {H}x{W} 256->256 res={res}: {N * (H // 4) * (W // 4) // 16 * 2} workgroups", file=sys.stderr, flush=True)
    K.conv2d_nhwc(x, w, b, padding=1, relu=1, out=y, residual=r, res_mode=1 if res else 0, winograd="f43")
    torch.cuda.synchronize()

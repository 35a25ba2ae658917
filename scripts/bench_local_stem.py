"""fused local-extractor stem vs the three separate kernels (GPU box only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
R = 256
x = torch.randn((R, 128, 128, 4), device=dev); x[..., 3] = 0
w1 = torch.randn((16, 3, 3, 4), device=dev) * 0.2; b1 = torch.randn((16,), device=dev)
w2 = torch.randn((32, 3, 3, 16), device=dev) * 0.1; b2 = torch.randn((32,), device=dev)
def timed(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def sep():
    t = K.conv2d_nhwc(x, w1, b1, padding=1, relu=1); t = K.conv2d_nhwc(t, w2, b2, padding=1, relu=1); return K.maxpool2d_nhwc(t, 2, 2, 0)
print(f"R={R}: separate {timed(sep):.3f} ms, fused {timed(lambda: K.local_stem_fused(x, w1, b1, w2, b2)):.3f} ms")

"""Run one conv shape a few times (for PMC profiling). usage: one_conv.py N H W Cin Cout k stride pad [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
N, H, W, Cin, Cout, k, s, p = [int(v) for v in sys.argv[1:9]]
it = int(sys.argv[9]) if len(sys.argv) > 9 else 3
dev = torch.device("cuda:0")
x = torch.randn((N, H, W, Cin), device=dev); w = K.prepare_conv_weights(torch.randn((Cout, k, k, Cin), device=dev) * 0.05, "all"); b = torch.randn((Cout,), device=dev)
y = K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1)
for _ in range(it):
    K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, out=y)
torch.cuda.synchronize()

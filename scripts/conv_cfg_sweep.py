"""Time one conv shape under every forced direct-kernel tile configuration (GLASS_CONV_CFG=1..6, read per call) and
the dispatcher's own choice.  usage: conv_cfg_sweep.py N H W Cin Cout k stride pad [res]   (GPU box only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
N, H, W, Cin, Cout, k, s, p = [int(v) for v in sys.argv[1:9]]
res = len(sys.argv) > 9 and sys.argv[9] == "res"
dev = torch.device("cuda:0")
x = torch.randn((N, H, W, Cin), device=dev); w = torch.randn((Cout, k, k, Cin), device=dev) * 0.05; b = torch.randn((Cout,), device=dev)
y = K.conv2d_nhwc(x, w, b, stride=s, padding=p, relu=1, winograd=False)
r = torch.randn_like(y) if res else None
kw = dict(stride=s, padding=p, relu=1, out=y, winograd=False, residual=r, res_mode=1 if res else 0)
flops = 2.0 * y.numel() * Cin * k * k
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    K.conv2d_nhwc(x, w, b, **kw)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    K.conv2d_nhwc(x, w, b, **kw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"cfg {os.environ.get('GLASS_CONV_CFG', 'auto'):5s} {ms:7.3f} ms  {flops / ms / 1e9:6.1f} TFLOP/s")

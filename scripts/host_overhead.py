"""Host-side cost of one K.conv2d_nhwc call (Python + ctypes + torch allocator), measured on tiny tensors."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
x = torch.randn((1, 8, 8, 64), device=dev); w = torch.randn((64, 3, 3, 64), device=dev); b = torch.randn((64,), device=dev)
w1 = torch.randn((64, 1, 1, 64), device=dev)
for name, fn in (("conv3x3 (winograd auto -> direct, tiny grid)", lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1)),
                 ("conv1x1", lambda: K.conv2d_nhwc(x, w1, b, relu=1)),
                 ("conv1x1 with out=", None),
                 ("torch.empty only", lambda: torch.empty((1, 8, 8, 64), device=dev)),
                 ("maxpool", lambda: K.maxpool2d_nhwc(x, 2, 2))):
    if fn is None:
        out = torch.empty((1, 8, 8, 64), device=dev)
        fn = lambda: K.conv2d_nhwc(x, w1, b, relu=1, out=out)
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5000
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:48s} host {1e6 * (t1 - t0) / n:6.1f} us/call   (+ drain {1e3 * (t2 - t1):.1f} ms)")

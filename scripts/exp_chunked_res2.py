"""Does running the res2 stage a few images at a time keep its 537 MB inter-layer tensors in the 256 MB Infinity Cache?
Three bottleneck blocks (1x1 256->64, 3x3 64->64, 1x1 64->256 + residual) on 8 x 256 x 256 maps, whole batch vs chunks."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
N, H, W = 8, 256, 256
g = torch.Generator().manual_seed(0)
mk = lambda co, k, ci: K.prepare_conv_weights((torch.randn((co, k, k, ci), generator=g) * (2.0 / (k * k * ci)) ** 0.5).to(dev), "fp32")
blocks = [(mk(64, 1, 256), mk(64, 3, 64), mk(256, 1, 64)) for _ in range(3)]
bias64, bias256 = torch.zeros((64,), device=dev), torch.zeros((256,), device=dev)
x0 = torch.randn((N, H, W, 256), device=dev)
bufs = {"a": torch.empty((N, H, W, 64), device=dev), "b": torch.empty((N, H, W, 64), device=dev),
        "y0": torch.empty((N, H, W, 256), device=dev), "y1": torch.empty((N, H, W, 256), device=dev)}


def stage(chunk):
    for n0 in range(0, N, chunk):
        sl = slice(n0, n0 + chunk)
        x = x0[sl]
        for i, (w1, w2, w3) in enumerate(blocks):
            a = K.conv2d_nhwc(x, w1, bias64, relu=1, out=bufs["a"][sl])
            b = K.conv2d_nhwc(a, w2, bias64, padding=1, relu=1, out=bufs["b"][sl])
            y = K.conv2d_nhwc(b, w3, bias256, relu=1, residual=x, res_mode=1, out=bufs["y%d" % (i & 1)][sl])
            x = y
    return x


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = stage(8).clone()
for chunk in (8, 4, 2, 1):
    out = stage(chunk)
    print(f"chunk {chunk}: {timeit(lambda: stage(chunk)):.3f} ms for the 3 blocks on 8 images; max diff vs whole batch {float((bufs['y0'] - ref).abs().max()):.1e}")

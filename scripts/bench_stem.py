"""Times the fused ResNet stem (csrc/backbone_stem.hip) against the two launches it replaces, and the last-column strip conv.
    python scripts/bench_stem.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)


def timeit(fn, n=30):
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


x = torch.zeros((8, 1024, 1024, 4))
x[..., :3] = torch.randn((8, 1024, 1024, 3), generator=g) * 60
x = x.to(dev)
w = torch.zeros((64, 7, 7, 4))
w[..., :3] = torch.randn((64, 7, 7, 3), generator=g) * 0.003
w, b = w.to(dev), torch.randn((64,), generator=g).to(dev)
off = K.default_routing().replace(stem=False)
t_f = timeit(lambda: K.backbone_stem_fused(x, w, b))
t_c = timeit(lambda: K.conv2d_nhwc(x, w, b, stride=2, padding=3, relu=1, routing=off))
y = K.conv2d_nhwc(x, w, b, stride=2, padding=3, relu=1, routing=off)
t_p = timeit(lambda: K.maxpool2d_nhwc(y, 3, 2, 1))
fl = 2.0 * 8 * 512 * 512 * 64 * 147
print(f"stem 8x1024x1024: fused {t_f:.3f} ms ({fl / t_f / 1e9:.1f} TF/s algorithmic) | conv {t_c:.3f} + pool {t_p:.3f} = {t_c + t_p:.3f} ms")
for R in (256, 128):
    xs = torch.randn((R, 16, 33, 256), generator=g).to(dev)
    ws = (torch.randn((256, 3, 3, 256), generator=g) * 0.02).to(dev)
    res = torch.randn((R, 16, 33, 256), generator=g).to(dev)
    cw = K.prepare_conv_weights(ws, ragged=True)
    t_split = timeit(lambda: K.conv2d_nhwc(xs, cw, b.repeat(4), padding=1, relu=1, residual=res, res_mode=1, winograd="f43"))
    t_whole = timeit(lambda: K.conv2d_nhwc(xs, cw, b.repeat(4), padding=1, relu=1, residual=res, res_mode=1, winograd="f43",
                                           routing=K.default_routing().replace(ragged=False)))
    out = torch.empty_like(xs)
    d = K.ConvDesc(R, 16, 33, 256, 256, 3, 3, 1, 1, 1, 1, 16, 33, 256, 256, 0, 1, 1, 1, 256)
    t_strip = timeit(lambda: K._last_column_strip(xs, cw.packs["col1"], b.repeat(4), res, out, d, 0))
    print(f"[{R},16,33,256]->256 +res: split {t_split:.3f} ms (strip alone {t_strip:.3f}) | whole-tile-column form {t_whole:.3f} ms")

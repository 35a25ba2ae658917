import os, sys, torch
sys.path.insert(0, "glass-text-spotting_amd")
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cases = [(8, 256, 256, 256, 256, 1, False), (256, 16, 33, 256, 256, 1, True), (256, 32, 32, 128, 128, 0, True), (256, 64, 64, 64, 64, 1, False),
         (8, 128, 128, 256, 256, 0, False), (256, 8, 32, 512, 256, 0, False), (8, 256, 256, 64, 64, 1, False), (3, 37, 45, 96, 384, 2, True), (256, 64, 64, 32, 64, 1, False)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rp = K.default_routing().replace(persist=True)
r0 = K.default_routing().replace(persist=False)
for N, H, W, Cin, Cout, relu, res in cases:
    x = torch.randn((N, H, W, Cin), generator=g).to(dev)
    w = K.prepare_conv_weights((torch.randn((Cout, 3, 3, Cin), generator=g) * 0.05).to(dev), ragged=True)
    b = torch.randn((Cout,), generator=g).to(dev)
    r = torch.randn((N, H, W, Cout), generator=g).to(dev) if res else None
    kw = dict(padding=1, relu=relu, residual=r, res_mode=1 if res else 0, winograd="f43")
    y0 = K.conv2d_nhwc(x, w, b, routing=r0, **kw)
    y1 = K.conv2d_nhwc(x, w, b, routing=rp, **kw)
    y2 = K.conv2d_nhwc(x, w, b, routing=rp, **kw)
    torch.cuda.synchronize()
    ctr = list(K._PERSIST_CTR.values())[0].cpu().tolist()
    eq = torch.equal(y0, y1) and torch.equal(y1, y2)
    t0 = timeit(lambda: K.conv2d_nhwc(x, w, b, routing=r0, **kw)); t1 = timeit(lambda: K.conv2d_nhwc(x, w, b, routing=rp, **kw))
    print(f"[{N},{H},{W},{Cin}]->{Cout} relu{relu} res{int(res)}: bit-equal {eq}, counters clean {sum(ctr) == 0}, one-shot {t0:.3f} ms, persistent {t1:.3f} ms ({t0 / t1:.3f}x)", flush=True)

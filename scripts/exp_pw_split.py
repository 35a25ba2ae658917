"""1x1 layers of the 8-image step: the fp32-MFMA kernels (pointwise / implicit GEMM, as routed) against the exact-product
bf16-split kernel (csrc/pointwise_split.hip) with nine and six piece products - us per launch (HIP events, back to back) and
max error / output range against float64.
    python scripts/exp_pw_split.py [--quick]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")


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
    return e0.elapsed_time(e1) / n * 1e3


# (N, H, W, Cin, Cout, stride, residual) - the 1x1 layers of profiles/r05_conv_table.txt with Cin % 32 == 0, Cout % 128 == 0
LAYERS = [(8, 256, 256, 256, 256, 1, 1), (8, 256, 256, 256, 512, 2, 0), (8, 128, 128, 512, 1024, 2, 0), (8, 128, 128, 512, 256, 1, 1),
          (8, 256, 256, 64, 256, 1, 1), (8, 64, 64, 1024, 2048, 2, 0), (8, 128, 128, 128, 512, 1, 1), (256, 8, 32, 512, 256, 1, 0),
          (8, 64, 64, 256, 1024, 1, 1), (8, 128, 128, 512, 128, 1, 0), (8, 32, 32, 512, 2048, 1, 1), (8, 64, 64, 1024, 256, 1, 0),
          (8, 32, 32, 2048, 512, 1, 0), (8, 32, 32, 2048, 256, 1, 0), (8, 256, 256, 256, 128, 2, 0), (256, 16, 33, 128, 256, 1, 0),
          (8192, 1, 1, 256, 2048, 1, 0), (800, 1, 1, 2048, 2048, 1, 0), (8192, 1, 1, 512, 256, 1, 0), (8192, 1, 1, 256, 256, 1, 0),
          (1, 64, 64, 256, 1024, 1, 1), (1, 256, 256, 256, 256, 1, 1), (1, 32, 32, 2048, 512, 1, 0)]
if "--quick" in sys.argv:
    LAYERS = LAYERS[:9]
torch.manual_seed(0)
tot = {"routed": 0.0, "pws9": 0.0, "pws6": 0.0}
for (N, H, W, Cin, Cout, st, has_res) in LAYERS:
    x = torch.randn((N, H, W, Cin), device=dev)
    w_raw = torch.randn((Cout, 1, 1, Cin), device=dev) * (1.0 / Cin ** 0.5)
    w = K.prepare_conv_weights(w_raw, "fp32")
    w.packs["pws"] = K.winograd_pack(w_raw, "pws")
    b = torch.randn((Cout,), device=dev)
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    res = torch.randn((N, Ho, Wo, Cout), device=dev) if has_res else None
    y = torch.empty((N, Ho, Wo, Cout), device=dev)
    kw = dict(stride=st, relu=1, residual=res, res_mode=1 if has_res else 0, out=y)
    # float64 reference on a sample of output pixels
    xs = x[:, ::st, ::st, :].reshape(-1, Cin)
    idx = torch.randint(0, xs.shape[0], (min(4096, xs.shape[0]),), device=dev)
    ref = xs[idx].double() @ w_raw.view(Cout, Cin).double().t() + b.double()
    if has_res:
        ref = ref + res.reshape(-1, Cout)[idx].double()
    ref = ref.clamp_min(0)
    rng = float(ref.abs().max())
    out = {}
    for name, force in (("routed", None), ("pws9", "pws9"), ("pws6", "pws6")):
        rt = K.default_routing().replace(split=0) if force is None else None      # "routed": what the fp32-MFMA routing picks
        t = timeit(lambda: K.conv2d_nhwc(x, w, b, winograd=force, routing=rt, **kw))
        path = K.last_conv_path()
        err = float((y.reshape(-1, Cout)[idx].double() - ref).abs().max()) / rng
        out[name] = (t, err, path)
        tot[name] += t
    gf = 2.0 * N * Ho * Wo * Cin * Cout / 1e9
    hbm = (x.numel() / (st * st) + y.numel() * (2 if has_res else 1)) * 4 / 1e6      # MB (strided reads: the pixels taken)
    print(f"[{N},{H},{W},{Cin}]->{Cout} s{st}{' +res' if has_res else ''}: " +
          "  ".join(f"{k} {v[0]:7.1f} us ({gf / v[0] * 1e3:5.1f} TF/s, err {v[1]:.1e}{', ' + v[2] if k == 'routed' else ''})" for k, v in out.items()) +
          f"  | {hbm:.0f} MB -> {hbm / out['pws9'][0]:.2f} TB/s")
print("total us: " + "  ".join(f"{k} {v:.0f}" for k, v in tot.items()))

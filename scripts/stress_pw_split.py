"""Is the bf16-split 1x1 kernel bit-stable beside other kernels?  Stream A runs it repeatedly, stream B one co-runner family at
a time; every result of A is compared with the result it gives alone.    python scripts/stress_pw_split.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, H, W, Cin, Cout = 8, 128, 128, 512, 256
x = torch.randn((N, H, W, Cin), device=dev)
w_raw = torch.randn((Cout, 1, 1, Cin), device=dev) * 0.05
w = K.prepare_conv_weights(w_raw, "fp32")
w.packs["pws"] = K.winograd_pack(w_raw, "pws")
b = torch.randn((Cout,), device=dev)
x3 = torch.randn((8, 128, 128, 256), device=dev)
w3 = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp32")
x2 = torch.randn((N, H, W, Cin), device=dev)
w2 = K.prepare_conv_weights(w_raw * 0.5, "fp32")
w2.packs["pws"] = K.winograd_pack(w_raw * 0.5, "pws")
e1 = torch.randn((64, 1 << 20), device=dev)
torch.cuda.synchronize()
force = sys.argv[1] if len(sys.argv) > 1 else "pws9"
ref = K.conv2d_nhwc(x, w, b, relu=1, winograd=force).clone()
refp = K.conv2d_nhwc(x, w, b, relu=1).clone()
print("path of the fp32 kernel:", K.last_conv_path(), " max |split - fp32| / max:", float((ref - refp).abs().max() / refp.abs().max()))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
corunners = {
    "nothing": lambda: None,
    "the split kernel itself": lambda: K.conv2d_nhwc(x2, w2, b, relu=1, winograd=force),
    "fp32 F(4x4) Winograd": lambda: K.conv2d_nhwc(x3, w3, None, padding=1, relu=1),
    "fp32 pointwise": lambda: K.conv2d_nhwc(x2, w2, b, relu=1, winograd=None, routing=K.default_routing().replace(split=0)),
    "torch elementwise (mul/add/relu)": lambda: torch.relu(e1 * 1.0001 + 0.5),
    "torch softmax": lambda: torch.softmax(e1, dim=1),
}
for name, co in corunners.items():
    bad, worst, where = 0, 0.0, None
    for it in range(60):
        with torch.cuda.stream(sb):
            for _ in range(3):
                co()
        with torch.cuda.stream(sa):
            y = K.conv2d_nhwc(x, w, b, relu=1, winograd=force)
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            worst = max(worst, float(d.max()))
            if where is None:
                idx = torch.nonzero(d.reshape(-1, Cout) > 0)
                where = (int(idx.shape[0]), idx[:6].tolist())
    print(f"beside {name}: {bad} of 60 runs differ" + (f"; max |diff| {worst:.3e}; first run: {where[0]} elements, (pixel, channel) {where[1]}" if bad else ""))

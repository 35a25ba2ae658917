"""Per-launch timing of glass_conv2d_nhwc inside the real pipeline (GPU box only)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd")); sys.path.insert(0, ROOT)
import torch
import glass_amd
from glass_amd.config import get_glass_cfg
from glass_amd.ops import native as K
from glass_amd.utils.synth import make_boxes, make_image, make_state_dict

dev = torch.device("cuda:0")
cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
model = glass_amd.build_model(cfg); model.load_state_dict(make_state_dict(1234))
B, S, R = 8, 1000, 32
inputs = [{"image": make_image(i, S, S).permute(2, 0, 1).float().contiguous().to(dev)} for i in range(B)]
boxes = [make_boxes(i, R, S, S).to(dev) for i in range(B)]
for _ in range(2):
    model.inference(inputs, override_boxes=boxes)
torch.cuda.synchronize()
orig = K.conv2d_nhwc
recs = []
def wrapped(x, w, bias=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig(x, w, bias, **kw); e1.record()
    recs.append((e0, e1, tuple(x.shape), tuple(w.shape), tuple(y.shape), kw.get("stride", 1), kw.get("res_mode", 0) if kw.get("residual") is not None else 0))
    return y
K.conv2d_nhwc = wrapped
model.inference(inputs, override_boxes=boxes)
torch.cuda.synchronize()
K.conv2d_nhwc = orig
agg = collections.OrderedDict()
tot = 0
for e0, e1, xs, ws, ys, st, rm in recs:
    ms = e0.elapsed_time(e1); tot += ms
    fl = 2.0 * ys[0] * ys[1] * ys[2] * ws[0] * ws[1] * ws[2] * ws[3]
    byts = 4.0 * (xs[0] * xs[1] * xs[2] * ws[3] + ys[0] * ys[1] * ys[2] * ws[0] * (2 if rm else 1))
    key = (xs, ws, st, rm)
    a = agg.setdefault(key, [0, 0.0, fl, byts]); a[0] += 1; a[1] += ms
print(f"total conv ms {tot:.2f}")
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (xs, ws, st, rm), (n, ms, fl, byts) in rows[:45]:
    print(f"{ms:7.2f} ms n={n:2d} {fl*n/ms/1e9:6.1f} TF/s {byts*n/ms/1e6:7.0f} GB/s(min) x={xs} w={ws} s={st} res={rm}")

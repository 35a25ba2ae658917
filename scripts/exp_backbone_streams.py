"""Experiment: backbone+RPN on the full batch (1 stream) vs two half batches on two streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd")); sys.path.insert(0, ROOT)
import torch
import glass_amd
from glass_amd.config import get_glass_cfg
from glass_amd.utils.synth import make_image, make_state_dict
dev = torch.device("cuda:0")
cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
m = glass_amd.build_model(cfg); m.load_state_dict(make_state_dict(1234))
inputs = [{"image": make_image(i, 1000, 1000).permute(2, 0, 1).float().contiguous().to(dev)} for i in range(8)]
il = m.preprocess_image(inputs)
x = il.nhwc4
hw = torch.tensor(il.image_sizes, dtype=torch.int32, device=dev)
pg = m.proposal_generator
def full():
    f = m.backbone.forward_nhwc(x)
    return pg.forward_batched([f[k] for k in pg.in_features], hw)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def halves():
    cur = torch.cuda.current_stream()
    outs = []
    for st, sl in ((s1, slice(0, 4)), (s2, slice(4, 8))):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            f = m.backbone.forward_nhwc(x[sl])
            outs.append(pg.forward_batched([f[k] for k in pg.in_features], hw[sl]))
    cur.wait_stream(s1); cur.wait_stream(s2)
    return outs
for fn in (full, halves, full, halves):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(fn.__name__, (time.perf_counter() - t0) / 10 * 1e3, "ms")

import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd")); sys.path.insert(0, ROOT)
import torch
import glass_amd
from glass_amd.config import get_glass_cfg
from glass_amd.postprocess import build_post_processor
from glass_amd.structures.core import Instances, RotatedBoxes
from glass_amd.utils.synth import make_boxes
cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
pp = build_post_processor(cfg)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def mk(i):
    inst = Instances((1000, 1000))
    inst.pred_boxes = RotatedBoxes(make_boxes(i, 32, 1000, 1000).to(dev))
    inst.scores = (torch.rand(32, generator=g) * 0.8 + 0.2).to(dev)
    inst.pred_classes = torch.zeros(32, dtype=torch.int64, device=dev)
    p = torch.softmax(torch.randn((32, 26, 97), generator=g) * 4, -1)
    inst.pred_text_prob = p.to(dev)
    return inst
for i in range(3): pp(mk(i))
n = 16
insts = [mk(100 + i) for i in range(n)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for inst in insts: out = pp(inst)
torch.cuda.synchronize(); print("device post-process ms/image:", (time.perf_counter() - t0) / n * 1e3, "kept", len(out))
insts = [mk(100 + i) for i in range(n)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for inst in insts: out = pp.host_call(inst)
torch.cuda.synchronize(); print("host restatement ms/image:", (time.perf_counter() - t0) / n * 1e3, "kept", len(out))
# batched: 8 images in one kernel
from glass_amd.ops import native as K
boxes = torch.stack([make_boxes(i, 32, 1000, 1000) for i in range(8)]).to(dev)
scores = (torch.rand((8, 32), generator=g) * 0.8 + 0.2).to(dev)
text = torch.softmax(torch.randn((8, 32, 26, 97), generator=g) * 4, -1).to(dev)
cnt = torch.full((8,), 32, dtype=torch.int32, device=dev)
for _ in range(2): pp.process_padded(boxes, scores, cnt, text, None, [(1000, 1000)] * 8)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): res = pp.process_padded(boxes, scores, cnt, text, None, [(1000, 1000)] * 8)
torch.cuda.synchronize(); print("batched device post-process ms/image:", (time.perf_counter() - t0) / 80 * 1e3, "kept", [len(r) for r in res])

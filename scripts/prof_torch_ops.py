"""Which torch (aten) operators still launch kernels inside one bench step, and from which source line.
    python scripts/prof_torch_ops.py [batch]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd")); sys.path.insert(0, ROOT)
import torch, glass_amd
from glass_amd.config import get_glass_cfg
from glass_amd.distributed import pack_words
from glass_amd.postprocess import build_post_processor
from glass_amd.modeling.fusion.recognizers_hybrid_head import prepare_injected_boxes
from glass_amd.utils.pipeline import drive
from glass_amd.utils.synth import make_boxes, make_image, make_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
model = glass_amd.build_model(cfg); model.load_state_dict(make_state_dict(1234))
post = build_post_processor(cfg)
images = [{"image": make_image(g, 1000, 1000).permute(2, 0, 1).float().contiguous().to(dev)} for g in range(B)]
boxes = prepare_injected_boxes([make_boxes(g, 32, 1000, 1000).to(dev) for g in range(B)], dev)


def step_g():
    out = yield from model.inference_g(images, override_boxes=boxes)
    det = out.batch
    words = yield from post.process_padded_g(det.boxes, det.scores, det.counts_dev, det.text, None, [(1000, 1000)] * B, {"orientations": det.orient})
    return pack_words(words.words, 100, 26)


for _ in range(3):
    drive(step_g())
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    drive(step_g())
    torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0 and not any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        where = next((s for s in (ev.stack or []) if "glass_amd" in s or "bench" in s or "scripts" in s), "?")
        rows[(ev.name, where.strip()[-110:])] += 1
for (name, where), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} {name:28s} {where}")
print("total aten ops with device time:", sum(rows.values()))

"""Capture the word post-processor's inputs of one bench step (the padded detections of 8 synthetic images) into
gpurun_out/pp_inputs.pt and time glass_postprocess_words on them (events around back-to-back launches)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import glass_amd
from glass_amd.config import get_glass_cfg
from glass_amd.ops import native as K
from glass_amd.postprocess import build_post_processor
from glass_amd.utils.pipeline import drive
from glass_amd.utils.synth import make_boxes, make_image, make_state_dict
dev = torch.device("cuda:0")
path = os.path.join(ROOT, "gpurun_out", "pp_inputs.pt")
cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
post = build_post_processor(cfg)
if not os.path.exists(path) or os.environ.get("PP_RECAPTURE"):
    model = glass_amd.build_model(cfg)
    model.load_state_dict(make_state_dict(1234))
    sets = []
    for s in range(3):
        inputs = [{"image": make_image(g + 1000 * s, 1000, 1000).permute(2, 0, 1).float().contiguous().to(dev)} for g in range(8)]
        boxes = [make_boxes(g + 1000 * s, 32, 1000, 1000).to(dev) for g in range(8)]
        det = drive(model.inference_g(inputs, override_boxes=boxes)).batch
        sets.append({k: getattr(det, k).cpu() for k in ("boxes", "scores", "counts_dev", "text")})
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(sets, path)
sets = torch.load(path)
thr, stop = post._thresholds(), None
print("thresholds", thr)
for s, d in enumerate(sets):
    b, sc, cnt, tx = (d[k].to(dev) for k in ("boxes", "scores", "counts_dev", "text"))
    stop = post.text_encoder.character.index("[s]")
    def run(): return K.postprocess_words(b, sc, cnt, tx, None, thr, stop)
    o = run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"set {s}: counts {cnt.tolist()} kept {o['count'].tolist()}  {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. output allocation)")

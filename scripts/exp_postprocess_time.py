import os, sys, torch
sys.path.insert(0, "glass-text-spotting_amd")
from glass_amd.ops import native as K
from glass_amd.utils.synth import make_boxes
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, KK, T, C = 8, 100, 26, 97
boxes = torch.zeros((N, KK, 5)); 
for i in range(N): boxes[i, :32] = make_boxes(i, 32, 1024, 1024)
boxes = boxes.to(dev)
scores = (torch.rand((N, KK), generator=g) * 0.8 + 0.2).to(dev)
text = torch.softmax(torch.randn((N, KK, T, C), generator=g) * 4, -1).to(dev)
thr = [2.0, 0.05, 0.5, 0.6, 0.5, 15.0, 0.1, 0.8]
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SIDE = int(os.environ.get("PP_SIDE", "1024")); OFF = int(os.environ.get("PP_SEED_OFF", "0"))
boxes.zero_()
for i in range(N): boxes[i, :32] = make_boxes(i + OFF, 32, SIDE, SIDE).to(dev)
for c in (0, 32, 100):
    cnt = torch.full((N,), c, dtype=torch.int32, device=dev)
    if c == 100:
        for i in range(N): boxes[i] = make_boxes(i, 100, 1024, 1024).to(dev)
    for tx in (None, text):
        o = K.postprocess_words(boxes, scores, cnt, tx, None, thr, C - 1)
        print(f"count {c} text {tx is not None}: {timeit(lambda: K.postprocess_words(boxes, scores, cnt, tx, None, thr, C - 1)):.1f} us  kept {o['count'].tolist()}")

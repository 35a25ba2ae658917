"""Experiment record (GPU box): the static-shape trunk of a step (GeneralizedRCNN._trunk: backbone + FPN + RPN + box head,
~100 launches) captured into ONE hipGraph vs launched kernel by kernel.   python scripts/exp_hipgraph_trunk.py H W [B]
Round 2, ROCm 7.2, MI355X, B=8 1000x1000: graph replay 1-3 ms SLOWER per step than the stream launches (bench with the
trunk graphed: 256.8 vs 264.0 images/s pipelined, 223.6 vs 239.8 one step at a time) - the runtime walks the graph node by
node with its own dependency tracking, while plain launches on one stream are already back to back (the step is
GPU-bound with 2 host cores, profiles/r02_bench_host_pinned_2cores.json).  Not built into the product for that reason."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd")); sys.path.insert(0, ROOT)
import torch, glass_amd
from glass_amd.config import get_glass_cfg
from glass_amd.ops import native as K
from glass_amd.utils.synth import make_image, make_state_dict
H, W = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
m = glass_amd.build_model(cfg); m.load_state_dict(make_state_dict(1234))
imgs = [make_image(300 + i, H, W).permute(2, 0, 1).float().contiguous().cuda() for i in range(B)]
batch = torch.zeros((B, (H + 31) // 32 * 32, (W + 31) // 32 * 32, 4), device="cuda:0")
hw = torch.tensor([[H, W]] * B, dtype=torch.int32).cuda()
with torch.no_grad():
    m.preprocess_image([{"image": im} for im in imgs], into=batch)
    t = m._trunk(batch, hw); torch.cuda.synchronize(); print("eager ok", t["oc"].tolist(), flush=True)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=cap):
        out = m._trunk(batch, hw)
    torch.cuda.synchronize(); print("capture ok", flush=True)
    g.replay(); torch.cuda.synchronize(); print("replay ok", out["oc"].tolist(), flush=True)
    for k in t:
        if not torch.equal(t[k], out[k]):
            print("DIFF", k, float((t[k].float() - out[k].float()).abs().max()))

    def timed(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    print(f"trunk {B} x {H}x{W}: eager {timed(lambda: m._trunk(batch, hw)):.3f} ms, graph replay {timed(g.replay):.3f} ms")

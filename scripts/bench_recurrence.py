"""BiLSTM recurrence (and decoder) timing: one launch per step vs one launch per layer, alone and next to a convolution stream.
    python scripts/bench_recurrence.py [R ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
Rs = [int(a) for a in sys.argv[1:]] or [32, 256]
T = 32


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn((8, 256, 256, 256), device=dev)
w = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp32")
yc = torch.empty((8, 256, 256, 256), device=dev)
side = torch.cuda.Stream(device=dev)
for R in Rs:
    g = torch.Generator().manual_seed(R)
    xg = (torch.randn((R, T, 2, 1024), generator=g) * 1.5).to(dev)
    whh = (torch.randn((2, 1024, 256), generator=g) * 0.08).to(dev)
    ref = K.bilstm_recurrence(xg, whh, 256, mode="steps")
    for mode in ("steps", (2, 1), (2, 2), (1, 1)):
        out = K.bilstm_recurrence(xg, whh, 256, mode=mode)
        same = torch.equal(out, ref)
        t_alone = timeit(lambda: K.bilstm_recurrence(xg, whh, 256, mode=mode))
        # next to a stream that keeps the chip full of F(4x4) convolution workgroups: time of the recurrence AND how many
        # convolutions completed meanwhile (the pipeline pays the chain's duration and the CUs it holds)
        torch.cuda.synchronize()
        stop = False
        n_conv = 0
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            for _ in range(40):
                K.conv2d_nhwc(x, w, None, padding=1, out=yc)
        e0.record()
        for _ in range(20):
            K.bilstm_recurrence(xg, whh, 256, mode=mode)
        e1.record()
        torch.cuda.synchronize()
        t_both = time.perf_counter() - t0
        t_loaded = e0.elapsed_time(e1) / 20 * 1e3
        print(f"R={R:5d} mode={str(mode):8s} bit-identical={same}  alone {t_alone:8.1f} us/layer   beside 40 convs: {t_loaded:8.1f} us/layer, "
              f"everything done in {t_both * 1e3:7.2f} ms")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(40):
            K.conv2d_nhwc(x, w, None, padding=1, out=yc)
    torch.cuda.synchronize()
    print(f"          (40 convs alone: {(time.perf_counter() - t0) * 1e3:7.2f} ms)")
# ---- decoder
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from glass_amd.config import get_glass_cfg
from glass_amd.modeling.recognition.recognizer_decoder import ASTER_V2
from glass_amd.structures.core import ShapeSpec
from glass_amd.utils.synth import make_state_dict
cfg = get_glass_cfg(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "glass_icdar15_mi355x.yaml"))
dec = ASTER_V2(cfg, ShapeSpec(channels=256))
dec.import_weights(make_state_dict(1234), dev, "roi_heads.recognizer_head.decoder.")
for R in Rs:
    g = torch.Generator().manual_seed(R)
    xd = torch.randn((R, 32, 256), generator=g).to(dev)
    ri = (torch.arange(R) // 32).to(torch.int32).to(dev)
    ni = int(ri.max()) + 1
    xp = K.linear(xd.view(R * 32, 256), dec.w["xW"], dec.w["xB"]).view(R, 32, 256)
    ref = K.attention_decode(xd, xp, dec.w, ri, ni, dec.num_classes, dec.max_word_len, 0, mode="steps")
    for mode in ("steps", (1, 1)):
        out = K.attention_decode(xd, xp, dec.w, ri, ni, dec.num_classes, dec.max_word_len, 0, mode=mode)
        err = float((out - ref).abs().max())
        t_alone = timeit(lambda: K.attention_decode(xd, xp, dec.w, ri, ni, dec.num_classes, dec.max_word_len, 0, mode=mode), n=20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            for _ in range(40):
                K.conv2d_nhwc(x, w, None, padding=1, out=yc)
        e0.record()
        for _ in range(10):
            K.attention_decode(xd, xp, dec.w, ri, ni, dec.num_classes, dec.max_word_len, 0, mode=mode)
        e1.record()
        torch.cuda.synchronize()
        print(f"decoder R={R:5d} mode={str(mode):8s} max |diff| vs steps {err:.2e}  alone {t_alone:8.1f} us   beside 40 convs: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us, "
              f"everything done in {(time.perf_counter() - t0) * 1e3:7.2f} ms")
print("status", K.recurrence_status())

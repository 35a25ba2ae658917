"""GPU: kernels stay bit-exact while ANOTHER HIP stream keeps the chip busy.

Round 3 root cause of the pipelined fp16-mode non-determinism (docs/DESIGN_history_r1-r3.md, "co-resident MFMA erratum"): on this MI355X /
ROCm 7.2 stack a packed-f32 VALU instruction whose LOW-result selector reads a high half (`v_pk_mul_f32 ... op_sel:[0,1]`,
what hipcc emits for `a.x * b.y` style float2 / float4 arithmetic) returns a wrong low half in lanes 48..63 when a
wavefront of another kernel issues a double-rate f16 / bf16 MFMA (v_mfma_f32_16x16x32_f16, 32x32x16_f16) on the same SIMD.
RoIAlign, the decoder, NMS and the post-processing kernels were built with such instructions; conv_h16_kernel of the other
in-flight step is the MFMA side.  The library is now built without packed-f32 selection (tests/test_isa_guard.py holds
that statically); these tests hold the behaviour."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hammer_streams():
    dev = torch.device("cuda:0")
    return torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)


def _run_beside(victim, hammer, iters):
    """`victim()` launched `iters` times on one stream while `hammer()` runs back to back on another; returns the outputs"""
    sv, sh = _hammer_streams()
    cur = torch.cuda.current_stream()
    sv.wait_stream(cur)
    sh.wait_stream(cur)
    outs = []
    for _ in range(iters):
        with torch.cuda.stream(sh):
            hammer()
        with torch.cuda.stream(sv):
            outs.append(victim())
    torch.cuda.synchronize()
    return outs


def test_roi_align_is_exact_beside_fp16_matrix_core_convolutions():
    """the reproducer that failed for 1000 of 1200 launches with the round-2 library"""
    from glass_amd.ops import native as K
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    img = torch.randn((1, 128, 160, 4), device=dev) * 50
    img[..., 3] = 0
    feat = torch.randn((1, 32, 40, 256), device=dev)
    boxes = torch.tensor([[60., 50., 40., 20., 30.], [100., 80., 50., 16., -20.], [40., 90., 30., 12., 75.], [80., 30., 60., 25., 5.]], device=dev)
    bidx = torch.zeros((4,), dtype=torch.int32, device=dev)
    hx = torch.randn((8, 64, 64, 256), device=dev)
    hw = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp16")

    def hammer():
        prev = K.set_conv_precision("fp16")
        try:
            K.conv2d_nhwc(hx, hw, None, padding=1)
        finally:
            K.set_conv_precision(prev)
    hammer()
    assert K.last_conv_path() == "packed_fp16"

    def crops():
        return K.roi_align_rotated([img], [1.0], boxes, bidx, (128, 128), 2, channels=4)

    def pooled():       # the recognizer pooler's shape: 256 channels, channel-interleaved output
        out = torch.zeros((4, 8, 32, 512), device=dev)
        return K.roi_align_rotated([feat], [0.25], boxes, bidx, (8, 32), 2, out=out, out_coff=1, out_cstride=2)

    for name, victim in (("image pooler", crops), ("recognizer pooler", pooled)):
        ref = victim().clone()
        torch.cuda.synchronize()
        bad = sum(0 if torch.equal(o, ref) else 1 for o in _run_beside(victim, hammer, 300))
        assert bad == 0, f"{name}: {bad} of 300 RoIAlign launches differ from the solo result while {beside} runs beside them"


# (label, precision, x shape, w shape, stride/pad, expected path): one case per fp16 kernel instantiation
_FP16_CASES = [
    ("conv_h16<16,1>", "fp16s", (8, 96, 96, 256), (256, 3, 3, 256), 1, "packed_fp16"),
    ("conv_h16<8,1>", "fp16s", (8, 64, 64, 256), (256, 3, 3, 256), 1, "packed_fp16"),
    ("conv_h16<4,1> (fc-like)", "fp16s", (800, 1, 1, 1024), (1024, 1, 1, 1024), 0, "packed_fp16"),
    ("conv_h16<8,2>", "fp16s", (8, 128, 128, 64), (64, 3, 3, 64), 1, "packed_fp16"),
    ("conv_h16<4,2>", "fp16s", (8, 64, 64, 64), (64, 3, 3, 64), 1, "packed_fp16"),
    ("conv_h16<2,2>", "fp16s", (1, 64, 64, 64), (64, 1, 1, 64), 0, "packed_fp16"),
    ("fp32 template, fp16 operands (Cin 32)", "fp16s", (8, 64, 64, 32), (64, 3, 3, 32), 1, "direct_fp16"),
    ("cast + conv_h16 (fp32 tensors, 'fp16' mode)", "fp16", (8, 64, 64, 128), (128, 3, 3, 128), 1, "packed_fp16"),
    ("fp32 template in 'fp16' mode (Cin 4)", "fp16", (4, 128, 128, 4), (16, 3, 3, 4), 1, "direct_fp16"),
]


@pytest.mark.parametrize("case", _FP16_CASES, ids=[c[0] for c in _FP16_CASES])
def test_fp16_kernels_are_deterministic_beside_the_fp32_winograd_kernel(case):
    """every fp16 kernel variant, 40 launches with conv3x3_wino43_f32 hammering from a second stream: bit-equal to solo"""
    from glass_amd.ops import native as K
    label, prec, xs, ws, pad, path = case
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    half_in = prec == "fp16s"
    x = torch.randn(xs, device=dev)
    if half_in:
        x = x.half()
    w = K.prepare_conv_weights(torch.randn(ws, device=dev) * 0.05, prec)
    b = torch.randn((ws[0],), device=dev)
    r = torch.randn(xs[:3] + (ws[0],), device=dev)
    if half_in:
        r = r.half()
    hx = torch.randn((8, 64, 64, 256), device=dev)
    hw = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp32")

    def victim():
        prev = K.set_conv_precision(prec)
        try:
            return K.conv2d_nhwc(x, w, b, padding=pad, relu=1, residual=r, res_mode=1)
        finally:
            K.set_conv_precision(prev)

    def hammer():
        K.conv2d_nhwc(hx, hw, None, padding=1, winograd="f43")

    ref = victim().clone()
    assert K.last_conv_path() == path, (label, K.last_conv_path())
    hammer()
    assert K.last_conv_path() == "winograd43"
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o, ref) else 1 for o in _run_beside(victim, hammer, 40))
    assert bad == 0, f"{label}: {bad} of 40 launches differ from the solo result"


def test_fused_fp16_stem_is_deterministic_beside_the_fp32_winograd_kernel():
    from glass_amd.ops import native as K
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    x = torch.randn((16, 128, 128, 4), device=dev)
    x[..., 3] = 0
    w1, b1 = torch.randn((16, 3, 3, 4), device=dev) * 0.2, torch.randn((16,), device=dev) * 0.1
    w2, b2 = torch.randn((32, 3, 3, 16), device=dev) * 0.1, torch.randn((32,), device=dev) * 0.1
    hx = torch.randn((8, 64, 64, 256), device=dev)
    hw = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp32")

    def victim():
        prev = K.set_conv_precision("fp16s")
        try:
            assert K.local_stem_supported(x, w1, w2)
            return K.local_stem_fused(x, w1, b1, w2, b2)
        finally:
            K.set_conv_precision(prev)
    ref = victim().clone()
    assert ref.dtype == torch.float16
    torch.cuda.synchronize()
    outs = _run_beside(victim, lambda: K.conv2d_nhwc(hx, hw, None, padding=1, winograd="f43"), 40)
    assert all(torch.equal(o, ref) for o in outs)


@pytest.mark.parametrize("beside", ["conv_h16_kernel", "conv1x1_pw_split"])
def test_fp32_kernels_are_exact_beside_fp16_matrix_core_convolutions(beside):
    """the other direction: the fp32 path's kernels (Winograd, implicit GEMM + residual, pointwise, GC attention) with
    conv_h16_kernel beside them - an fp32 step pipelined next to an fp16-mode step must not change by a bit - and with the
    bf16-split 1x1 kernel beside them, which is what two pipelined fp32 steps do to each other since round 5"""
    from glass_amd.ops import native as K
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    hx = torch.randn((8, 64, 64, 256), device=dev)
    hw = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp16")
    hw1 = K.prepare_conv_weights(torch.randn((256, 1, 1, 256), device=dev) * 0.05, "fp32")

    def hammer():
        if beside == "conv1x1_pw_split":
            K.conv2d_nhwc(hx, hw1, None, winograd="pws9")
            return
        prev = K.set_conv_precision("fp16")
        try:
            K.conv2d_nhwc(hx, hw, None, padding=1)
        finally:
            K.set_conv_precision(prev)

    x = torch.randn((4, 32, 40, 256), device=dev)
    w3 = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp32")
    w1 = K.prepare_conv_weights(torch.randn((64, 1, 1, 256), device=dev) * 0.05, "fp32")
    wp = K.prepare_conv_weights(torch.randn((256, 1, 1, 256), device=dev) * 0.05, "fp32")
    b = torch.randn((256,), device=dev)
    r = torch.randn((4, 32, 40, 256), device=dev)
    xg = torch.randn((8, 8 * 32, 512), device=dev).view(8, 8, 32, 512)
    # the GLASS shapes of glass_gc_attention_inplace: C 512, 8 heads, P 256, HW 256
    gw = {"w_mask": torch.randn((64,), device=dev) * 0.05, "b_mask": torch.zeros((1,), device=dev),
          "w1": torch.randn((256, 512), device=dev) * 0.05, "b1": torch.zeros((256,), device=dev),
          "ln_g": torch.ones((256,), device=dev), "ln_b": torch.zeros((256,), device=dev),
          "w2": torch.randn((512, 256), device=dev) * 0.05, "b2": torch.zeros((512,), device=dev)}
    victims = {
        "winograd": lambda: K.conv2d_nhwc(x, w3, b, padding=1, relu=1, residual=r, res_mode=1, winograd=True),
        "implicit GEMM": lambda: K.conv2d_nhwc(x, w3, b, padding=1, relu=1, residual=r, res_mode=1, winograd=False),
        "1x1 (64 ch)": lambda: K.conv2d_nhwc(x, w1, None),
        "pointwise": lambda: (K.set_pointwise("all"), K.conv2d_nhwc(x, wp, b, relu=1, routing=K.default_routing().replace(split=0)), K.set_pointwise(True))[1],
        "pointwise split (bf16 MFMA)": lambda: K.conv2d_nhwc(x, wp, b, relu=1, winograd="pws9"),
        "gc attention": lambda: K.gc_attention_inplace(xg.clone(), 8, gw["w_mask"], gw["b_mask"], gw["w1"], gw["b1"], gw["ln_g"],
                                                       gw["ln_b"], gw["w2"], gw["b2"]),
    }
    for name, victim in victims.items():
        ref = victim().clone()
        torch.cuda.synchronize()
        bad = sum(0 if torch.equal(o, ref) else 1 for o in _run_beside(victim, hammer, 60))
        assert bad == 0, f"{name}: {bad} of 60 launches differ from the solo result while {beside} runs beside them"

"""GPU: the BASELINE.json configurations as parity / stress cases.
configs[0] 512x512, pretrain-style cfg (stock GeneralizedRCNN) vs the CPU oracle end to end;
configs[1] backbone+FPN at 1000x1000 (padded 1024^2) vs the CPU oracle (full size, B=1);
configs[4] TextOCR-style stress shape (1333 long side, 100 RoIs/img, orientation head off): fp32 B=8 stress (runs, finite,
           well-formed, parity on one image) and the config's own precision - fp16 storage ('fp16s') and fp16 operands
           ('fp16') - against the oracle EMULATING that arithmetic (oracle/glass_cpu.py: emulate()).
(Collection order of the -m gpu suite is parity first: a_stages, b_configs, c_mask_branch, d_known_answers, e_host_tail, then
the kernel sweeps f_ops, then y_properties / z_pipeline / z_stress - a late flake must not hide the reference-golden tests.)"""
import os

import numpy as np
import pytest
import torch

from parity import (TOL, assert_detections_close, assert_same_box_set as _assert_same_box_set, assert_text_prob_close, maxdiff,
                    teacher_forced_text_probs, text_prob_stats)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(opts=()):
    from glass_amd.config import get_glass_cfg
    return get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"] + list(opts))


@pytest.fixture(scope="module")
def sd():
    from glass_amd.utils.synth import make_state_dict
    return make_state_dict(1234)


def _compare_proposals(det, n, ref_props, what):
    """RPN output of image n (padded device tensors on det.proposals) vs the oracle's (boxes, logits)"""
    pb, pl, pc = det.proposals
    c = int(pc[n])
    rb, rs = ref_props
    assert c == len(rb), f"{what}: {c} proposals vs oracle {len(rb)}"
    d = maxdiff(pl[n, :c].cpu().numpy(), rs.numpy())
    print(f"[parity] {what}: {c} proposals, max |dlogit| = {d:.3e}")
    assert d < TOL
    _assert_same_box_set(pb[n, :c].cpu().numpy(), rb.numpy())


def _compare_detections(det, n, ref, what, tol=TOL, box_atol=2e-3):
    c = det.counts_host[n]
    got = {"scores": det.scores[n, :c].cpu().numpy(), "boxes": det.boxes[n, :c].cpu().numpy(),
           "orientations": None if det.orient is None else det.orient[n, :c].cpu().numpy(),
           "kept": None if det.kept_index is None else det.kept_index[n, :c].cpu().numpy()}
    refd = {"scores": ref["scores"].numpy(), "pred_boxes": ref["pred_boxes"].numpy(),
            "orientations": None if ref.get("orientations") is None else ref["orientations"].numpy(),
            "kept": None if ref.get("kept") is None else ref["kept"].numpy()}
    assert_detections_close(got, refd, tol=tol, what=what + " detections", box_atol=box_atol)


def test_config0_512_pretrain_style_end_to_end_vs_oracle(sd):
    import glass_amd
    from glass_amd.utils.synth import make_image
    from oracle import glass_cpu as O
    cfg = _cfg(["MODEL.META_ARCHITECTURE", "GeneralizedRCNN"])
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    img = make_image(20, 512, 512).permute(2, 0, 1).float()
    res = m.inference([{"image": img.cuda()}], do_postprocess=False)
    out, det = res[0], res.batch
    ref = O.glass_inference(sd, [img], cfg)[0]
    assert len(out) > 0
    _compare_proposals(det, 0, ref["proposals"], "configs[0] 512x512")
    _compare_detections(det, 0, ref, "configs[0] 512x512")
    assert_text_prob_close(out.pred_text_prob.cpu().numpy(), ref["pred_text_prob"].numpy(), what="configs[0] 512x512 text")


def test_config1_backbone_fpn_full_size_vs_oracle(sd):
    import glass_amd
    from glass_amd.utils.synth import make_image
    from oracle import glass_cpu as O
    cfg = _cfg()
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    img = make_image(21, 1000, 1000).permute(2, 0, 1).float()
    il = m.preprocess_image([{"image": img.cuda()}])
    assert tuple(il.nhwc4.shape) == (1, 1024, 1024, 4)
    feats = m.backbone.forward_nhwc(il.nhwc4)
    x, _ = O.preprocess([img], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
    ref = O.resnet50_fpn(sd, x)
    for k, r in ref.items():
        got = feats[k].permute(0, 3, 1, 2).cpu()
        assert got.shape == r.shape
        scale = max(1.0, float(r.abs().max()))
        assert float((got - r).abs().max()) < 1e-3 * scale, k


def test_config4_textocr_stress_shape_runs(sd):
    import glass_amd
    from glass_amd.utils.synth import make_boxes, make_image
    cfg = _cfg(["MODEL.ORIENTATION_ON", False])
    sd2 = {k: v for k, v in sd.items() if "orientation_pred" not in k}
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd2)
    B, H, W, R = 8, 1000, 1333, 100
    inputs = [{"image": make_image(30 + i, H, W).permute(2, 0, 1).float().contiguous().cuda()} for i in range(B)]
    boxes = [make_boxes(30 + i, R, H, W).cuda() for i in range(B)]
    out = m.inference(inputs, override_boxes=boxes)
    assert len(out) == B
    for o in out:
        inst = o["instances"]
        assert inst.image_size == (H, W) and 0 < len(inst) <= R
        assert torch.isfinite(inst.pred_text_prob).all() and inst.pred_text_prob.shape[1:] == (26, 97)
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() < 200 * 2 ** 30


def test_config4_textocr_one_image_vs_oracle(sd):
    """one 1000x1333 image of the TextOCR shape (orientation head off) against the CPU oracle: character
    probabilities of 12 injected word boxes (the detection stages are compared at this size by config 1)."""
    import glass_amd
    from glass_amd.utils.synth import make_boxes, make_image
    from oracle import glass_cpu as O
    cfg = _cfg(["MODEL.ORIENTATION_ON", False])
    sd2 = {k: v for k, v in sd.items() if "orientation_pred" not in k}
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd2)
    H, W = 1000, 1333
    img = make_image(61, H, W).permute(2, 0, 1).float().contiguous()
    boxes = [make_boxes(61, 12, H, W)]
    out = m.inference([{"image": img.cuda()}], do_postprocess=False, override_boxes=[boxes[0].cuda()])[0]
    ref = O.glass_inference(sd2, [img], cfg, injected_boxes=boxes)[0]
    p, q = out.pred_text_prob.cpu().numpy(), ref["pred_text_prob"].numpy()
    assert p.shape == q.shape == (12, 26, 97)
    assert_text_prob_close(p, q, what="configs[4] 1000x1333 one image, 12 RoIs, text")


def test_config2_bench_workload_one_image_vs_oracle(sd):
    """configs[2] (the metric's workload) for ONE 1000x1000 image: real RPN + box head, recognition of 32 injected word
    boxes (exactly what bench.py runs 8 of per step), against the CPU oracle - detections and character probabilities."""
    import glass_amd
    from glass_amd.utils.synth import make_boxes, make_image
    from oracle import glass_cpu as O
    cfg = _cfg()
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    img = make_image(0, 1000, 1000).permute(2, 0, 1).float().contiguous()          # bench image 0 (seed 1000 + 0)
    boxes = [make_boxes(0, 32, 1000, 1000)]                                          # bench boxes 0 (seed 2000 + 0)
    ref = O.glass_inference(sd, [img], cfg, injected_boxes=boxes)[0]
    res = m.inference([{"image": img.cuda()}], do_postprocess=False, override_boxes=[boxes[0].cuda()])
    det = res.batch
    p, q = det.text.cpu().numpy(), ref["pred_text_prob"].numpy()
    assert p.shape == q.shape == (32, 26, 97)
    assert_text_prob_close(p, q, what="configs[2] 1000x1000, 32 RoIs, text")
    # the RPN and the box head ran teacher-free on the image: same proposals (set, scores), same detections (kept
    # proposal indices, scores, boxes, orientations) as the oracle
    assert len(ref["proposals"][0]) > 0
    _compare_proposals(det, 0, ref["proposals"], "configs[2] 1000x1000")
    _compare_detections(det.detected, 0, ref, "configs[2] 1000x1000")


def test_config2_bench_batch_of_8_vs_oracle_image_by_image(sd):
    """the TIMED dispatch.  bench.py's step is B = 8 images with 32 injected boxes each (input set 0: image and
    box seeds g = 0..7) - [8,256,256,256] FPN convs, [256,16,33,256] local-extractor convs, whose conv routing (wide / narrow
    shape, tile counts, XCD map, pointwise rule) depends on the batch's tile count.  The reference's contract is per-image
    results (glass_runner.py:93-96): every image of the batch against `O.glass_inference` of that image alone -
    proposals, detections, character probabilities."""
    import glass_amd
    from glass_amd.utils.synth import make_boxes, make_image
    from oracle import glass_cpu as O
    cfg = _cfg()
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    B, R = 8, 32
    imgs = [make_image(g, 1000, 1000).permute(2, 0, 1).float().contiguous() for g in range(B)]
    boxes = [make_boxes(g, R, 1000, 1000) for g in range(B)]
    res = m.inference([{"image": im.cuda()} for im in imgs], do_postprocess=False, override_boxes=[b.cuda() for b in boxes])
    det = res.batch
    text = det.text.cpu().numpy()
    assert text.shape == (B * R, 26, 97)
    worst = 0.0
    for g in range(B):
        ref = O.glass_inference(sd, [imgs[g]], cfg, injected_boxes=[boxes[g]])[0]
        worst = max(worst, assert_text_prob_close(text[g * R:(g + 1) * R], ref["pred_text_prob"].numpy(),
                                                  what=f"configs[2] B=8 image {g}, 32 RoIs, text"))
        _compare_proposals(det, g, ref["proposals"], f"configs[2] B=8 image {g}")
        _compare_detections(det.detected, g, ref, f"configs[2] B=8 image {g}")
    print(f"[parity] configs[2] at the bench batch (B=8, R=256): max |dp| over the 8 images = {worst:.3e}")


def test_config1_backbone_fpn_batch_of_8_vs_oracle(sd):
    """configs[1] as quoted - bs = 8 at 1000 x 1000 - every FPN level of every image against the oracle's
    ResNet-50 + FPN of that image alone."""
    import glass_amd
    from glass_amd.utils.synth import make_image
    from oracle import glass_cpu as O
    cfg = _cfg()
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    B = 8
    imgs = [make_image(g, 1000, 1000).permute(2, 0, 1).float().contiguous() for g in range(B)]
    il = m.preprocess_image([{"image": im.cuda()} for im in imgs])
    assert tuple(il.nhwc4.shape) == (B, 1024, 1024, 4)
    feats = {k: v.permute(0, 3, 1, 2).cpu() for k, v in m.backbone.forward_nhwc(il.nhwc4).items()}
    worst = {}
    for g in range(B):
        x, _ = O.preprocess([imgs[g]], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
        ref = O.resnet50_fpn(sd, x)
        for k, r in ref.items():
            got = feats[k][g:g + 1]
            assert got.shape == r.shape
            rel = float((got - r).abs().max()) / max(1.0, float(r.abs().max()))
            worst[k] = max(worst.get(k, 0.0), rel)
            assert rel < 1e-3, (g, k, rel)
    print("[parity] configs[1] B=8 backbone+FPN, max |d| / range per level over the 8 images: " +
          ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))


def test_local_extractor_at_the_bench_batch_vs_oracle(sd):
    """the local extractor alone at R = 256 crops (8 images x 32 RoIs: the [256,16,33,256] / [256,32,32,128] /
    [256,64,64,64] launches of the timed step) against `O.local_extractor`, itself pinned on the reference module
    (tests/golden/local_extractor.npz; reference local_feature_extraction.py:95-188)."""
    import glass_amd
    from oracle import glass_cpu as O
    cfg = _cfg()
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    R = 256
    g = torch.Generator().manual_seed(4242)
    x = (torch.rand((R, 3, 128, 128), generator=g) * 255.0 - 115.0)
    x = torch.nn.functional.avg_pool2d(x, 3, 1, 1)                 # crops of a smooth image, mean-subtracted range
    got = m.roi_heads.hybrid_net.forward(x.cuda()).cpu()
    worst = 0.0
    for lo in range(0, R, 64):                                     # the oracle in chunks of 64 crops (memory, not semantics)
        ref = O.local_extractor(sd, x[lo:lo + 64])
        assert got[lo:lo + 64].shape == ref.shape == (64, 256, 8, 32)
        worst = max(worst, float((got[lo:lo + 64] - ref).abs().max()) / max(1.0, float(ref.abs().max())))
    print(f"[parity] local extractor at R = 256: max |d| / range = {worst:.3e}")
    assert worst < 1e-3


def test_trained_like_dynamic_range_at_the_bench_batch_vs_oracle(sd):
    """parity must survive a real checkpoint, not only the O(1) activations of seed 1234.  The checkpoint is
    re-scaled (`synth.widen_dynamic_range`: res4 / res5 / FPN / local layer3-4 conv gains x1.5-2.5, BN gamma up to x3 with
    running_var down to 0.1 -> 3x3 layers read inputs of up to 150 ... 420, pyramid levels reach 300 ... 440) and the images are
    text-like (dark flat background, bright anti-aliased strokes: `synth.make_text_image`), B = 8 at 1000 x 1000 with 32 injected
    boxes per image - the bench's dispatch, so res2 ... res4, the FPN / RPN convs and the local extractor run the F(4x4,3x3)
    kernel.  North-star bound (1e-3) on proposal logits, detection scores / boxes and character probabilities, image by image.
    For the log, the same batch under `f43 = False` (F(2x2,3x3)) and `winograd = False` (direct fp32 MFMA): how much of the delta
    is the transform and how much is fp32 summation order."""
    import glass_amd
    from glass_amd.utils.synth import make_boxes, make_text_image, widen_dynamic_range
    from oracle import glass_cpu as O
    cfg = _cfg()
    wsd = widen_dynamic_range(sd)
    B, R = 8, 32
    imgs = [make_text_image(g, 1000, 1000).permute(2, 0, 1).float().contiguous() for g in range(B)]
    boxes = [make_boxes(100 + g, R, 1000, 1000) for g in range(B)]
    refs = [O.glass_inference(wsd, [imgs[g]], cfg, injected_boxes=[boxes[g]])[0] for g in range(B)]
    peak = max(float(r["proposals"][1].abs().max()) for r in refs)
    print(f"[parity] trained-like checkpoint: oracle proposal logits up to {peak:.1f}")
    inputs = [{"image": im.cuda()} for im in imgs]
    dboxes = [b.cuda() for b in boxes]
    for label, switches in (("default routing (F(4x4,3x3) where it pays)", {}), ("f43 off (F(2x2,3x3))", {"f43": False}),
                            ("winograd off (direct fp32 MFMA)", {"winograd": False})):
        m = glass_amd.build_model(cfg)
        for k, v in switches.items():
            setattr(m.routing, k, v)                     # this model's own Routing; stamped on its weights at load
        m.load_state_dict(wsd)
        det = m.inference(inputs, do_postprocess=False, override_boxes=dboxes).batch
        text = det.text.cpu().numpy()
        pb, pl, pc = det.proposals
        dl = dp = 0.0
        for g in range(B):
            c = int(pc[g])
            if c == len(refs[g]["proposals"][0]):
                dl = max(dl, maxdiff(np.sort(pl[g, :c].cpu().numpy()), np.sort(refs[g]["proposals"][1].numpy())))
            q = refs[g]["pred_text_prob"].numpy()
            live = q.sum(-1) > 0
            dp = max(dp, float(np.abs(text[g * R:(g + 1) * R] - q)[live][:, None].max()) if live.any() else 0.0)
        print(f"[parity] trained-like checkpoint, {label}: max |dlogit| (sorted proposals) = {dl:.3e}, raw max |dp| text = {dp:.3e}")
        if switches:
            continue
        for g in range(B):
            assert_text_prob_close(text[g * R:(g + 1) * R], refs[g]["pred_text_prob"].numpy(),
                                   what=f"trained-like checkpoint, image {g}, text")
            _compare_proposals(det, g, refs[g]["proposals"], f"trained-like checkpoint, image {g}")
            _compare_detections(det.detected, g, refs[g], f"trained-like checkpoint, image {g}")


def test_config4_fp16_conv_mode_tracks_the_fp32_path(sd):
    """BASELINE configs[4] asks for an fp16 run: MODEL.CONV_PRECISION fp16 routes every conv / linear through
    glass_conv2d_nhwc_f16 (operands rounded to fp16, fp16 MFMA, fp32 accumulate and storage).  The op itself is exact
    against conv(fp16(x), fp16(w)) (tests/test_gpu_f_ops.py); here the whole model in that mode stays close to the fp32
    path on the TextOCR-style cfg: same recognised characters for > 90 % of the steps, mean probability delta < 5e-3."""
    import glass_amd
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import make_boxes, make_image
    sd2 = {k: v for k, v in sd.items() if "orientation_pred" not in k}
    H, W = 480, 640
    img = make_image(62, H, W).permute(2, 0, 1).float().contiguous().cuda()
    boxes = [make_boxes(62, 16, H, W).cuda()]
    outs = {}
    for prec in ("fp32", "fp16"):
        cfg = _cfg(["MODEL.ORIENTATION_ON", False, "MODEL.CONV_PRECISION", prec])
        m = glass_amd.build_model(cfg)
        m.load_state_dict(sd2)
        res = m.inference([{"image": img}], do_postprocess=False, override_boxes=boxes)
        outs[prec] = (res.batch.text.cpu().numpy(), K.last_conv_path())
        assert K.conv_precision() == "fp32"                       # the model restores the global setting
    assert outs["fp16"][1] in ("direct_fp16", "packed_fp16") and outs["fp32"][1] in ("direct", "winograd")
    p, q = outs["fp16"][0], outs["fp32"][0]
    live = q.sum(-1) > 0
    # greedy decoding: one flipped character re-routes the rest of that word, so the bound is on agreement and on the
    # mean, not on the maximum
    assert (p.argmax(-1)[live] == q.argmax(-1)[live]).mean() > 0.9
    assert np.abs(p - q).mean() < 5e-3


def _compare_reduced(det, n, ref, what):
    """fp16 modes: proposals and detections matched as SETS (order / thresholds move under fp16 noise)"""
    from parity import match_box_sets
    pb, pl, pc = det.proposals
    c = int(pc[n])
    fr, fg, ds, db = match_box_sets(pb[n, :c].cpu().numpy(), pl[n, :c].cpu().numpy(), ref["proposals"][0].numpy(), ref["proposals"][1].numpy())
    print(f"[parity] {what}: proposals {c} vs oracle {len(ref['proposals'][0])}: matched {fr:.3f} / {fg:.3f}, max |dlogit| {ds:.3e}, max |dbox| {db:.3e}")
    assert fr >= 0.98 and fg >= 0.98 and ds < 0.06 and db < 1.0       # measured: >= 0.99 matched, dlogit <= 4.1e-2, dbox <= 0.51 px
    d = det.detected if det.detected is not None else det
    k = d.counts_host[n]
    fr, fg, ds, db = match_box_sets(d.boxes[n, :k].cpu().numpy(), d.scores[n, :k].cpu().numpy(), ref["pred_boxes"].numpy(), ref["scores"].numpy())
    print(f"[parity] {what}: detections {k} vs oracle {len(ref['scores'])}: matched {fr:.3f} / {fg:.3f}, max |dscore| {ds:.3e}, max |dbox| {db:.3e}")
    # (a detection is scored on ITS proposal, which moved by up to db pixels: the score bound is the proposal logits' bound;
    # observed over the conv kernels this mode has had, i.e. over fp32 summation orders: 5e-3 ... 7e-2)
    # measured: all but at most ONE detection of 8-22 matched (17 / 18 = 0.944 is the worst case), dscore <= 6.8e-2, dbox <= 0.89 px
    assert fr >= 0.94 and fg >= 0.94 and ds < 0.09 and db < 1.5       # px / degrees on boxes of up to ~1000 px


@pytest.mark.parametrize("prec", ["fp16", "fp16s"])
def test_fp16_modes_match_their_emulating_oracle_small(sd, prec):
    """MODEL.CONV_PRECISION fp16 (operands rounded, fp32 storage) and fp16s (fp16 STORAGE on the conv path) against the
    oracle run in the SAME arithmetic (oracle.glass_cpu.emulate) on a 2-image batch: proposals and detections as sets
    (teacher-free), character probabilities on injected boxes.  What differs between the two sides is fp32 summation
    order, which every fp16 rounding downstream can amplify to one fp16 ulp (1e-3 relative) on single activations."""
    import glass_amd
    from glass_amd.utils.synth import make_boxes, make_image
    from oracle import glass_cpu as O
    cfg = _cfg(["MODEL.CONV_PRECISION", prec])
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    sizes = [(120, 150), (128, 100)]
    imgs = [make_image(70 + i, h, w).permute(2, 0, 1).float() for i, (h, w) in enumerate(sizes)]
    boxes = [make_boxes(70 + i, 6, h, w) * torch.tensor([1, 1, 0.4, 0.5, 1.0]) for i, (h, w) in enumerate(sizes)]
    with O.emulate(prec):
        ref = O.glass_inference(sd, imgs, cfg, injected_boxes=boxes)
    res = m.inference([{"image": im.cuda()} for im in imgs], do_postprocess=False, override_boxes=[b.cuda() for b in boxes])
    det = res.batch
    # every step of every RoI: the product's decoder fed the emulating oracle's symbols (no near-tie cut)
    flat = torch.cat(boxes).contiguous().cuda()
    ri = torch.tensor([0] * 6 + [1] * 6, dtype=torch.int32, device="cuda:0")
    enc = _encoder_output(m, [{"image": im.cuda()} for im in imgs], flat, ri, 2)
    qq = np.concatenate([r["pred_text_prob"].numpy() for r in ref], 0)
    tf = teacher_forced_text_probs(m.roi_heads.recognizer_head.decoder, enc, qq)
    mx, mean, _, agree = text_prob_stats(tf, qq, f"{prec} small batch, teacher-forced vs the emulating oracle")
    # measured (round 6, 312 live steps): fp16 max 2.9e-3 / mean 9.1e-6 / agreement 1.0; fp16s 2.3e-3 / 1.3e-5 / 1.0
    assert mx < 6e-3 and mean < 5e-5 and agree > 0.99
    for n, r in enumerate(ref):
        _compare_reduced(det, n, r, f"{prec} image {n}")
        # free-running greedy decoding under fp16 noise: steps whose top-2 gap is below 1e-2 may legitimately pick the other
        # character; RoIs are compared up to their first such step, and the share of RoIs that have one is the ORACLE's own
        # property (random weights give flat distributions), asserted at what the oracle says instead of a blanket 1.0
        q = r["pred_text_prob"].numpy()
        srt = np.sort(q, axis=-1)
        tied = float(((((srt[..., -1] - srt[..., -2]) < 1e-2) & (q.sum(-1) > 0)).any(1)).mean())
        assert_text_prob_close(res[n].pred_text_prob.cpu().numpy(), q, tol=4e-3, tie_eps=1e-2,
                               what=f"{prec} image {n} text (6 injected boxes, free-running)", max_tied=min(1.0, tied + 1e-9))


def _encoder_output(m, inputs, boxes_flat, roi_image, n_images):
    """the product's own decoder input for these boxes: pyramid -> poolers -> local extractor -> fusion -> CNN -> BiLSTM"""
    il = m.preprocess_image(inputs)
    feats = m.backbone.forward_nhwc(il.nhwc4)
    _, inter = m.roi_heads.recognizer_branch_batched(il.nhwc4, feats, boxes_flat, roi_image, n_images, return_intermediates=True)
    head = m.roi_heads.recognizer_head
    return head.encoder.forward_nhwc(head.backbone.forward_nhwc(inter["fused"]))


def test_config4_fp16_storage_full_shape_vs_emulating_oracle(sd):
    """BASELINE configs[4] in its stated precision: one 1000 x 1333 image of the TextOCR shape (orientation head off),
    100 injected RoIs, fp16 STORAGE on the conv path.  Three comparisons, each with an ASSERTED bound:
      (a) TEACHER-FORCED against the oracle emulating exactly that arithmetic: the product's decoder fed the oracle's previous
          symbols - every live step of every RoI compared, no near-tie cut (`max_tied`);
      (b) TEACHER-FORCED against the FP32 oracle - the reference's arithmetic; the reference has no fp16 path, so the 1e-3
          north-star bound does not apply to this mode and THIS is its stated distance: arg-max agreement, mean and 95th
          percentile of |dp|, at the values measured;
      (c) free-running (what the mode ships) against the emulating oracle, RoIs cut at their first near-tie as before - now
          with the measured share of cut RoIs asserted instead of 1.0 - and proposals / detections as sets."""
    import glass_amd
    from glass_amd.utils.synth import make_boxes, make_image
    from oracle import glass_cpu as O
    cfg = _cfg(["MODEL.ORIENTATION_ON", False, "MODEL.CONV_PRECISION", "fp16s"])
    sd2 = {k: v for k, v in sd.items() if "orientation_pred" not in k}
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd2)
    H, W, R = 1000, 1333, 100
    img = make_image(61, H, W).permute(2, 0, 1).float().contiguous()
    boxes = [make_boxes(61, R, H, W)]
    inputs = [{"image": img.cuda()}]
    res = m.inference(inputs, do_postprocess=False, override_boxes=[boxes[0].cuda()])
    with O.emulate("fp16s"):
        ref = O.glass_inference(sd2, [img], cfg, injected_boxes=boxes)[0]
    ref32 = O.glass_inference(sd2, [img], cfg, injected_boxes=boxes)[0]["pred_text_prob"].numpy()
    det = res.batch
    p, q = det.text.cpu().numpy(), ref["pred_text_prob"].numpy()
    assert p.shape == q.shape == ref32.shape == (R, 26, 97)
    enc = _encoder_output(m, inputs, boxes[0].cuda().contiguous(), torch.zeros((R,), dtype=torch.int32, device="cuda:0"), 1)
    dec = m.roi_heads.recognizer_head.decoder
    # (a) every step of every RoI against the emulating oracle
    tf = teacher_forced_text_probs(dec, enc, q)
    mx, mean, p95, agree = text_prob_stats(tf, q, "configs[4] fp16 storage, teacher-forced vs the emulating oracle")
    assert mx < FP16S_TF_EMU_MAX and mean < FP16S_TF_EMU_MEAN and agree > FP16S_TF_EMU_AGREE
    # (b) ... and against the fp32 oracle (the reference's arithmetic)
    tf32 = teacher_forced_text_probs(dec, enc, ref32)
    mx, mean, p95, agree = text_prob_stats(tf32, ref32, "configs[4] fp16 storage, teacher-forced vs the FP32 oracle")
    assert mean < FP16S_TF_F32_MEAN and p95 < FP16S_TF_F32_P95 and mx < FP16S_TF_F32_MAX and agree > FP16S_TF_F32_AGREE
    # (c) free-running
    srt = np.sort(q, axis=-1)
    tied = float(((((srt[..., -1] - srt[..., -2]) < 1e-2) & (q.sum(-1) > 0)).any(1)).mean())
    print(f"[parity] configs[4] fp16 storage: share of RoIs whose oracle meets a near-tie (< 1e-2) at some step: {tied:.2f}")
    assert_text_prob_close(p, q, tol=8e-3, tie_eps=1e-2, what="configs[4] fp16 storage, 1000x1333, 100 RoIs, text (free-running)",
                           max_tied=min(1.0, tied + 0.05))
    _compare_reduced(det, 0, ref, "configs[4] fp16 storage 1000x1333")


# measured on MI355X (round 6, 2600 live (RoI, step) pairs; `pytest -s` prints the values next to these bounds; bounds = ~2-3 x):
#   vs the emulating oracle: max |dp| 5.5e-3, mean 1.5e-5, p95 of the per-step max 1.5e-3, arg-max agreement 0.9992
#   vs the FP32 oracle:      max |dp| 6.2e-3, mean 2.5e-5, p95 2.4e-3, arg-max agreement 0.9965 (9 of 2600 steps read another character)
FP16S_TF_EMU_MAX, FP16S_TF_EMU_MEAN, FP16S_TF_EMU_AGREE = 1.2e-2, 5e-5, 0.995
FP16S_TF_F32_MAX, FP16S_TF_F32_MEAN, FP16S_TF_F32_P95, FP16S_TF_F32_AGREE = 1.5e-2, 8e-5, 5e-3, 0.99


def _fp16_ulps(got, ref):
    """|got - ref| in units of the fp16 ulp at |ref| (2^-10 relative; 2^-24 absolute below the smallest normal)"""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    e = np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14)))
    return np.abs(got - ref) / 2.0 ** (e - 10)


def _hist(u):
    return {f"<={k}": round(float((u <= k).mean()), 5) for k in (0, 1, 2, 4, 8, 32)} | {"max": round(float(u.max()), 1)}


def test_fp16s_stages_teacher_forced_ulp_histograms(sd):
    """why the END-TO-END fp16s bounds are 1e-2-class while every kernel is exact to summation order.
    Stage by stage, the product in fp16-storage mode against the oracle emulating that arithmetic, errors in fp16 ulps:
      * one conv fed identical inputs: the two sides differ by fp32 summation order only -> the stored fp16 value is
        identical for 99.9 % of the elements and one ulp off for the rest (fp32 sums that sit on a rounding boundary; a few
        cancelling sums are off by more ulps of their own tiny value);
      * the backbone + FPN (53 + 8 layers, each re-rounding to fp16): flips compound - median 1-2 ulps of the value, about
        one and a half fp16 epsilons (1.6e-3) of the range at worst, level by level;
      * the recognition branch fed the ORACLE's pyramid (teacher-forced): crops exact, local / global / fused features
        within 1e-3 of the range, character probabilities within 2e-3.
    The 1e-2 / set-matching bounds of the end-to-end fp16s tests are what the chained discontinuities (top-k, NMS, greedy
    arg-max) make of ulp-level differences, not kernel error; the teacher-forced bounds asserted here are the tight ones."""
    import glass_amd
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import make_boxes, make_image
    from oracle import glass_cpu as O
    dev = torch.device("cuda:0")
    cfg = _cfg(["MODEL.CONV_PRECISION", "fp16s"])
    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    sizes = [(192, 256), (160, 224)]
    imgs = [make_image(80 + i, h, w).permute(2, 0, 1).float() for i, (h, w) in enumerate(sizes)]
    boxes = [make_boxes(80 + i, 5, h, w) * torch.tensor([1, 1, 0.4, 0.5, 1.0]) for i, (h, w) in enumerate(sizes)]
    with O.emulate("fp16s"):
        x, _ = O.preprocess(imgs, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 32)
        feats_ref = O.resnet50_fpn(sd, x)
        probs_ref, inter = O.recognizer_branch(sd, x, feats_ref, boxes, cfg, return_intermediates=True)
    prev = K.set_conv_precision("fp16s")
    try:
        # (1) one layer, identical inputs: the stem
        nhwc4 = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 1)).contiguous().to(dev)
        w = m.backbone.w["stem"]
        y = K.conv2d_nhwc(nhwc4, *w, stride=2, padding=3, relu=1, out_dtype=torch.float16)
        with O.emulate("fp16s"):
            y_ref = O._st(O._cbn(x, sd, "backbone.bottom_up.stem.conv1", stride=2, padding=3, relu=True))
        yn = y.float().permute(0, 3, 1, 2).cpu().numpy()
        u = _fp16_ulps(yn, y_ref.numpy())
        d = float(np.abs(yn - y_ref.numpy()).max() / float(y_ref.abs().max()))
        print(f"[fp16s ulps] stem conv, identical input: {_hist(u)}, max |d| / range {d:.2e}")
        # (more than one ulp only where the sum cancels: |result| << |terms|, so an fp32-rounding-sized difference of the
        #  sum spans several ulps of the tiny result)
        assert (u <= 1).mean() > 0.9999 and (u > 0).mean() < 0.02 and d < 2.0 ** -10          # one fp16 ulp of the largest value
        # (2) the whole backbone + FPN
        feats = m.backbone.forward_nhwc(nhwc4)
        for k in ("p2", "p3", "p4", "p5"):
            assert feats[k].dtype == torch.float16
            u = _fp16_ulps(feats[k].float().permute(0, 3, 1, 2).cpu().numpy(), feats_ref[k].numpy())
            rng = float(feats_ref[k].abs().max())
            d = float(np.abs(feats[k].float().permute(0, 3, 1, 2).cpu().numpy() - feats_ref[k].numpy()).max())
            print(f"[fp16s ulps] backbone + FPN {k}: {_hist(u)}, max |d| / range {d / rng:.2e}")
            assert np.median(u) <= 2 and d / rng < 4e-3      # (signed, cancelling FPN sums: ulps of the VALUE overstate; see d / range)
        # (3) recognition branch on the ORACLE's pyramid (fp16-representable by construction)
        tf = {k: v.permute(0, 2, 3, 1).contiguous().to(dev).half() for k, v in feats_ref.items()}
        for k, v in tf.items():
            assert torch.equal(v.float().cpu(), feats_ref[k].permute(0, 2, 3, 1))      # nothing lost by the cast
        bcat = torch.cat(boxes).contiguous().to(dev)
        ri = torch.tensor([0] * 5 + [1] * 5, dtype=torch.int32, device=dev)
        probs, got = m.roi_heads.recognizer_branch_batched(nhwc4, tf, bcat, ri, 2, return_intermediates=True)
    finally:
        K.set_conv_precision(prev)
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(float(np.abs(np.asarray(b)).max()), 1e-30))
    xcat = got["xcat"].permute(0, 3, 1, 2).cpu().numpy()
    r = {"crops": rel(got["crops"][..., :3].permute(0, 3, 1, 2).cpu().numpy(), inter["crops"].numpy()),
         "local": rel(xcat[:, 0::2], inter["local"].numpy()), "global": rel(xcat[:, 1::2], inter["global"].numpy()),
         "fused": rel(got["fused"].permute(0, 3, 1, 2).cpu().numpy(), inter["fused"].numpy())}
    dp = float(np.abs(probs.cpu().numpy() - probs_ref.numpy()).max())
    print(f"[fp16s teacher-forced] recognition branch, max |d| / range: {r}, character probabilities max |dp| {dp:.2e}")
    # measured (round 3): crops 5.9e-6, global 5.0e-4, local 1.2e-3, fused 8.3e-5 of the range; probabilities 2.6e-4
    assert r["crops"] < 1e-5 and r["global"] < 1e-3 and r["local"] < 2.5e-3 and r["fused"] < 5e-4
    # the decoder too is teacher-forced (fed the oracle's symbols): every live step of every RoI, no near-tie cut
    head = m.roi_heads.recognizer_head
    enc = head.encoder.forward_nhwc(head.backbone.forward_nhwc(got["fused"]))
    tf = teacher_forced_text_probs(head.decoder, enc, probs_ref.numpy())
    mx, mean, _, agree = text_prob_stats(tf, probs_ref.numpy(), "fp16s teacher-forced pyramid AND decoder vs the emulating oracle")
    assert mx < 1e-3 and mean < 1e-5 and agree > 0.99          # measured (round 6, 260 live steps): 2.8e-4 / 1.9e-6 / 1.0

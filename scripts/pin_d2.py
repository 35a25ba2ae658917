"""Pin the detectron2-owned half of the oracle on detectron2 ITSELF - the one-command job for the first user who has one.

The reference reaches these ops through detectron2 v0.6 (an un-vendored, un-installable dependency here: no network, no
detectron2 wheel in either image), so `oracle/d2_ops.c` + `oracle/d2ops.py` restate its published algorithms from memory
(`[d2-recall]`) and are held to analytic known answers only (tests/known_answers.py,
tests/test_oracle_d2ops.py).  A mis-remembered convention that is self-consistent - `>=` vs `>` in NMS, the ordering inside
`find_top_rrpn_proposals`, the half-pixel shift of ROIAlignRotated - is invisible to both sides of our parity tests.

THIS SCRIPT CANNOT RUN IN THE BUILD CONTAINER OR ON THE GPU BOX (import detectron2 fails in both).  Where detectron2 (v0.6,
CPU build is enough) IS importable:

    python scripts/pin_d2.py                 # runs d2's own ops on seeded inputs, compares with oracle/d2ops.py,
                                             # writes tests/golden/d2_<op>.npz (inputs + detectron2's outputs)
    python -m pytest tests/test_oracle_d2_pins.py     # from then on: the oracle is held to those files, everywhere

    python scripts/pin_d2.py --dry-run       # (works anywhere) lists every op it would pin and the reference call site it serves

Reference call sites of the ops (amazon-science/glass-text-spotting): ROIAlignRotated / ROIPooler
glass/modeling/fusion/recognizers_hybrid_head.py:188-205,320,550,556; nms_rotated / batched_nms_rotated / pairwise_iou_rotated
glass/modeling/roi_heads/rotated_fast_rcnn.py:131, glass/postprocess/post_processor_rotated_boxes.py:120,181,
glass/postprocess/post_processor_academic.py:73,112; Box2BoxTransformRotated rotated_fast_rcnn.py:342 (via d2
FastRCNNOutputLayers) and the RRPN; RotatedAnchorGenerator + find_top_rrpn_proposals via PROPOSAL_GENERATOR "RRPN"
(configs/glass_finetune_icdar15.yaml); RotatedBoxes.clip / scale post_processor_academic.py:118-178.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------------------------- seeded inputs
def _boxes(seed: int, n: int, H: int = 512, W: int = 640) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    u = torch.rand((n, 5), generator=g)
    return torch.stack([u[:, 0] * W, u[:, 1] * H, 4 + u[:, 2] * 200, 3 + u[:, 3] * 80, u[:, 4] * 360 - 180], 1).float()


def _pairs():
    from known_answers import random_box_pairs              # the 8 families of tests/known_answers.py (thin, shared edge, ...)
    b1, b2, fam = random_box_pairs(4000, 7)
    return torch.from_numpy(b1), torch.from_numpy(b2), fam


def _feat(seed: int, N: int, C: int, H: int, W: int) -> torch.Tensor:
    return torch.randn((N, C, H, W), generator=torch.Generator().manual_seed(seed))


def _nms_scene(seed: int, n: int):
    """dense overlapping boxes with TIED scores and IoUs sitting exactly on the threshold: what separates `>` from `>=` and a
    stable from an unstable sort"""
    b = _boxes(seed, n, 200, 200)
    b[n // 2:] = b[: n - n // 2] + torch.tensor([3.0, 1.0, 2.0, 0.5, 2.0])
    s = torch.rand((n,), generator=torch.Generator().manual_seed(seed + 1))
    s[::5] = 0.5                                             # ties
    b[1] = torch.tensor([0.0, 0.0, 1.0, 1.0, 0.0]); b[2] = torch.tensor([0.5, 0.0, 1.0, 1.0, 0.0])     # IoU = 1/3 exactly
    s[1], s[2] = 0.99, 0.98
    return b, s


# ----------------------------------------------------------------------------------------------- the pins
# name -> (what d2 entry point, Appendix row, inputs(), run_d2(inputs), run_oracle(inputs), (rtol, atol))
def _pins():
    from oracle import d2ops
    P = {}

    def pin(name, d2_entry, appendix, inputs, run_d2, run_oracle, tol=(1e-5, 1e-5)):
        P[name] = dict(d2=d2_entry, appendix=appendix, inputs=inputs, run_d2=run_d2, run_oracle=run_oracle, tol=tol)

    # ---- A9 rotated IoU
    def iou_in():
        b1, b2, fam = _pairs()
        return {"b1": b1, "b2": b2, "family": torch.from_numpy(fam)}

    def iou_d2(i):
        from detectron2.layers.rotated_boxes import pairwise_iou_rotated
        return {"iou_diag": pairwise_iou_rotated(i["b1"], i["b2"]).diagonal() if len(i["b1"]) <= 512 else
                torch.cat([pairwise_iou_rotated(i["b1"][k:k + 500], i["b2"][k:k + 500]).diagonal() for k in range(0, len(i["b1"]), 500)])}

    def iou_or(i):
        return {"iou_diag": torch.cat([d2ops.pairwise_iou_rotated(i["b1"][k:k + 500], i["b2"][k:k + 500]).diagonal()
                                       for k in range(0, len(i["b1"]), 500)])}
    pin("pairwise_iou_rotated", "detectron2.layers.rotated_boxes.pairwise_iou_rotated (box_iou_rotated_utils.h)", "A9", iou_in, iou_d2, iou_or,
        (1e-4, 2e-5))

    # ---- A8 NMS
    def nms_in():
        b, s = _nms_scene(11, 300)
        return {"boxes": b, "scores": s, "idxs": torch.arange(300) % 3, "thresholds": torch.tensor([0.1, 1.0 / 3.0, 0.35, 0.5, 0.99])}

    def nms_d2(i):
        from detectron2.layers import batched_nms_rotated, nms_rotated
        out = {}
        for k, t in enumerate(i["thresholds"].tolist()):
            out[f"keep_{k}"] = nms_rotated(i["boxes"], i["scores"], t)
            out[f"bkeep_{k}"] = batched_nms_rotated(i["boxes"], i["scores"], i["idxs"], t)
        return out

    def nms_or(i):
        out = {}
        for k, t in enumerate(i["thresholds"].tolist()):
            out[f"keep_{k}"] = d2ops.nms_rotated(i["boxes"], i["scores"], t)
            out[f"bkeep_{k}"] = d2ops.batched_nms_rotated(i["boxes"], i["scores"], i["idxs"], t)
        return out
    pin("nms_rotated", "detectron2.layers.nms_rotated / batched_nms_rotated (CPU op: iou >= threshold suppresses? order of equal scores?)", "A8",
        nms_in, nms_d2, nms_or, (0, 0))

    # ---- A11 ROIAlignRotated: the three poolers' geometries (box 7x7 sr 2; recognizer 8x32 sr 0 adaptive; image 128x128 sr 2 on 3 ch)
    def ra_in():
        feats = _feat(3, 2, 8, 64, 80)
        b = _boxes(5, 40, 256, 320)
        b[0] = torch.tensor([100.0, 80.0, 60.0, 20.0, 0.0])       # axis-aligned (angle 0 == aligned RoIAlign)
        b[1] = torch.tensor([100.0, 80.0, 60.0, 20.0, 90.0])
        b[2] = torch.tensor([-20.0, 300.0, 80.0, 30.0, 33.0])     # mostly outside the map
        b[3] = torch.tensor([10.0, 10.0, 0.5, 0.5, 10.0])         # tiny (roi size clamps? d2's rotated op does NOT clamp to 1)
        rois = torch.cat([(torch.arange(40) % 2).float()[:, None], b], 1)
        return {"x": feats, "rois": rois}

    CASES = (("box", (7, 7), 0.25, 2), ("rec", (8, 32), 0.25, 0), ("img", (16, 16), 1.0, 2))

    def ra_d2(i):
        from detectron2.layers import ROIAlignRotated
        return {n: ROIAlignRotated(sz, sc, sr)(i["x"], i["rois"]) for n, sz, sc, sr in CASES}

    def ra_or(i):
        return {n: d2ops.roi_align_rotated(i["x"], i["rois"], sz, sc, sr) for n, sz, sc, sr in CASES}
    pin("roi_align_rotated", "detectron2.layers.ROIAlignRotated (ROIAlignRotated_cpu.cpp)", "A11", ra_in, ra_d2, ra_or, (1e-4, 1e-5))

    # ---- A10 ROIPooler level assignment + pooled output over 5 levels
    def rp_in():
        feats = {f"f{l}": _feat(20 + l, 2, 4, 256 >> l, 320 >> l) for l in range(5)}       # strides 4..64 of a 1024 x 1280 image
        b0, b1 = _boxes(31, 25, 1024, 1280), _boxes(32, 17, 1024, 1280)
        b0[:6, 2:4] = torch.tensor([[16.0, 16.0], [111.9, 112.1], [112.0, 112.0], [224.0, 224.0], [448.0, 448.0], [2000.0, 900.0]])   # level edges
        return {**feats, "boxes0": b0, "boxes1": b1}

    def rp_d2(i):
        from detectron2.modeling.poolers import ROIPooler, assign_boxes_to_levels
        from detectron2.structures import RotatedBoxes
        feats = [i[f"f{l}"] for l in range(5)]
        scales = [1.0 / (4 << l) for l in range(5)]
        pooler = ROIPooler(output_size=(7, 7), scales=scales, sampling_ratio=2, pooler_type="ROIAlignRotated")
        boxes = [RotatedBoxes(i["boxes0"]), RotatedBoxes(i["boxes1"])]
        return {"levels": assign_boxes_to_levels(boxes, 2, 6, 224, 4), "pooled": pooler(feats, boxes)}

    def rp_or(i):
        feats = [i[f"f{l}"] for l in range(5)]
        scales = [1.0 / (4 << l) for l in range(5)]
        return {"levels": d2ops.assign_boxes_to_levels(torch.cat([i["boxes0"], i["boxes1"]]), 2, 6),
                "pooled": d2ops.roi_pooler(feats, scales, [i["boxes0"], i["boxes1"]], (7, 7), 2)}
    pin("roi_pooler", "detectron2.modeling.poolers.ROIPooler(pooler_type='ROIAlignRotated') + assign_boxes_to_levels", "A10", rp_in, rp_d2, rp_or,
        (1e-4, 1e-5))

    # ---- A5 RotatedAnchorGenerator (the shipped cfg: sizes 16..256, 3 aspect ratios, 4 angles -> 12 anchors per cell)
    SIZES, RATIOS, ANGLES, STRIDES = [[16], [32], [64], [128], [256]], [[0.25, 0.5, 1.0]] * 5, [[-90, -45, 0, 45]] * 5, [4, 8, 16, 32, 64]

    def ag_in():
        return {"hw": torch.tensor([[24, 32], [12, 16], [6, 8], [3, 4], [2, 2]])}

    def ag_d2(i):
        from detectron2.layers import ShapeSpec
        from detectron2.modeling.anchor_generator import RotatedAnchorGenerator
        gen = RotatedAnchorGenerator(sizes=SIZES, aspect_ratios=RATIOS, angles=ANGLES, strides=STRIDES, offset=0.0)
        feats = [torch.zeros((1, 1, int(h), int(w))) for h, w in i["hw"].tolist()]
        return {f"anchors{l}": a.tensor for l, a in enumerate(gen(feats))}

    def ag_or(i):
        out = {}
        for l, (h, w) in enumerate(i["hw"].tolist()):
            cell = d2ops.rotated_cell_anchors(SIZES[l][0], RATIOS[l], ANGLES[l])
            out[f"anchors{l}"] = d2ops.rotated_grid_anchors(int(h), int(w), STRIDES[l], cell, 0.0)
        return out
    pin("rotated_anchor_generator", "detectron2.modeling.anchor_generator.RotatedAnchorGenerator (cell anchors + grid order)", "A5", ag_in, ag_d2, ag_or,
        (1e-6, 1e-5))

    # ---- A6 Box2BoxTransformRotated.apply_deltas (RPN weights (1,1,1,1,1), box head weights (10,10,5,5,1); clamp; angle wrap)
    def bt_in():
        g = torch.Generator().manual_seed(41)
        d = torch.randn((200, 5), generator=g) * torch.tensor([0.5, 0.5, 0.4, 0.4, 0.6])
        d[0] = 0.0
        d[1] = torch.tensor([0.0, 0.0, 9.0, 9.0, 0.0])            # past the scale clamp log(1000/16)
        d[2] = torch.tensor([0.0, 0.0, 0.0, 0.0, 4.0])            # angle wraps past +-180
        return {"deltas": d, "boxes": _boxes(42, 200)}

    def bt_d2(i):
        from detectron2.modeling.box_regression import Box2BoxTransformRotated
        return {"rpn": Box2BoxTransformRotated(weights=(1.0, 1.0, 1.0, 1.0, 1.0)).apply_deltas(i["deltas"], i["boxes"]),
                "box_head": Box2BoxTransformRotated(weights=(10.0, 10.0, 5.0, 5.0, 1.0)).apply_deltas(i["deltas"], i["boxes"])}

    def bt_or(i):
        return {"rpn": d2ops.apply_deltas_rotated(i["deltas"], i["boxes"], (1.0, 1.0, 1.0, 1.0, 1.0)),
                "box_head": d2ops.apply_deltas_rotated(i["deltas"], i["boxes"], (10.0, 10.0, 5.0, 5.0, 1.0))}
    pin("box2box_transform_rotated", "detectron2.modeling.box_regression.Box2BoxTransformRotated.apply_deltas", "A6", bt_in, bt_d2, bt_or, (1e-5, 1e-4))

    # ---- A13 RotatedBoxes.clip / scale / nonempty
    def rb_in():
        b = _boxes(51, 120, 300, 400)
        b[:20, 4] = torch.linspace(-1.5, 1.5, 20)                 # around the clip_angle_threshold of 1 degree
        b[20:30, 4] = torch.tensor([179.5, -179.5, 180.0, -180.0, 181.0, 359.0, -359.0, 90.0, -90.0, 0.0])
        return {"boxes": b, "hw": torch.tensor([300, 400]), "scale": torch.tensor([1.7, 0.6])}

    def rb_d2(i):
        from detectron2.structures import RotatedBoxes
        c = RotatedBoxes(i["boxes"].clone()); c.clip(tuple(i["hw"].tolist()))
        s = RotatedBoxes(i["boxes"].clone()); s.scale(float(i["scale"][0]), float(i["scale"][1]))
        return {"clipped": c.tensor, "scaled": s.tensor, "nonempty": c.nonempty().to(torch.int64)}

    def rb_or(i):
        import math
        c = d2ops.clip_rotated_(i["boxes"].clone(), tuple(i["hw"].tolist()))
        b = i["boxes"].clone()
        sx, sy = float(i["scale"][0]), float(i["scale"][1])
        th = b[:, 4] * math.pi / 180.0
        co, si = torch.cos(th), torch.sin(th)
        b[:, 0] *= sx; b[:, 1] *= sy
        b[:, 2] *= torch.sqrt((sx * co) ** 2 + (sy * si) ** 2)
        b[:, 3] *= torch.sqrt((sx * si) ** 2 + (sy * co) ** 2)
        b[:, 4] = torch.atan2(sx * si, sy * co) * 180 / math.pi
        return {"clipped": c, "scaled": b, "nonempty": ((c[:, 2] > 0) & (c[:, 3] > 0)).to(torch.int64)}
    pin("rotated_boxes_clip_scale", "detectron2.structures.RotatedBoxes.clip / .scale / .nonempty", "A13", rb_in, rb_d2, rb_or, (1e-5, 1e-4))

    # ---- A7 find_top_rrpn_proposals (per-level top-k with ties, finite filter, clip, min size, batched NMS across levels, post top-k)
    def fp_in():
        g = torch.Generator().manual_seed(61)
        out = {}
        for l, n in enumerate((3000, 800, 200)):
            p = torch.stack([_boxes(70 + 10 * l + k, n, 300, 400) for k in range(2)], 0)
            lg = torch.randn((2, n), generator=g)
            lg[:, ::7] = lg[:, 3:4]                                # tied logits: the sort's tie order decides who makes the cut
            p[0, 5, 2] = float("inf"); lg[1, 9] = float("nan")     # non-finite entries are dropped before NMS
            out[f"props{l}"], out[f"logits{l}"] = p, lg
        return out

    ARGS = dict(nms_thresh=0.7, pre_nms_topk=1000, post_nms_topk=300, min_box_size=0.0)

    def fp_d2(i):
        from detectron2.modeling.proposal_generator.rrpn import find_top_rrpn_proposals
        from detectron2.structures import ImageList
        res = find_top_rrpn_proposals([i[f"props{l}"] for l in range(3)], [i[f"logits{l}"] for l in range(3)], [(300, 400), (280, 400)],
                                      ARGS["nms_thresh"], ARGS["pre_nms_topk"], ARGS["post_nms_topk"], ARGS["min_box_size"], training=False)
        out = {}
        for n, r in enumerate(res):
            out[f"boxes{n}"], out[f"logits{n}"] = r.proposal_boxes.tensor, r.objectness_logits
        return out

    def fp_or(i):
        res = d2ops.find_top_rrpn_proposals([i[f"props{l}"] for l in range(3)], [i[f"logits{l}"] for l in range(3)], [(300, 400), (280, 400)], **ARGS)
        out = {}
        for n, (b, s) in enumerate(res):
            out[f"boxes{n}"], out[f"logits{n}"] = b, s
        return out
    pin("find_top_rrpn_proposals", "detectron2.modeling.proposal_generator.rrpn.find_top_rrpn_proposals", "A7", fp_in, fp_d2, fp_or, (1e-5, 1e-4))
    return P


def compare(name, got, ref, tol):
    """-> list of mismatch strings (empty = the oracle reproduces detectron2 on this op)"""
    bad = []
    for k in ref:
        a, b = np.asarray(got[k]), np.asarray(ref[k])
        if a.shape != b.shape:
            bad.append(f"{name}.{k}: shape {a.shape} vs detectron2 {b.shape}")
        elif a.dtype.kind in "iub" or b.dtype.kind in "iub":
            if not np.array_equal(a, b):
                bad.append(f"{name}.{k}: {int((a != b).sum())} of {a.size} integers differ from detectron2")
        elif not np.allclose(a, b, rtol=tol[0], atol=tol[1], equal_nan=True):
            bad.append(f"{name}.{k}: max |d| {float(np.nanmax(np.abs(a - b))):.3e} vs detectron2 (rtol {tol[0]}, atol {tol[1]})")
    return bad


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--dry-run", action="store_true", help="list the ops that would be pinned; needs no detectron2")
    ap.add_argument("--out", default=GOLDEN)
    args = ap.parse_args()
    pins = _pins()
    if args.dry_run:
        for name, p in pins.items():
            ins = p["inputs"]()
            print(f"{name:28s} Appendix {p['appendix']:4s} {p['d2']}\n{'':28s} -> {os.path.relpath(os.path.join(args.out, 'd2_' + name + '.npz'), ROOT)}; "
                  f"inputs: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in ins.items()))
        print(f"{len(pins)} ops; run without --dry-run where `import detectron2` works (v0.6; CPU build suffices)")
        return 0
    try:
        import detectron2                                             # noqa: F401
    except Exception as e:                                            # noqa: BLE001
        print(f"detectron2 is not importable here ({type(e).__name__}: {e}).\nNothing was written - the d2-owned half of the oracle stays "
              f"'parity unpinned' (DESIGN.md section 4).  `--dry-run` lists what this script pins.", file=sys.stderr)
        return 2
    print(f"detectron2 {getattr(detectron2, '__version__', '?')}: pinning {len(pins)} ops")
    failures = []
    for name, p in pins.items():
        ins = p["inputs"]()
        try:
            ref = {k: v.detach().cpu() for k, v in p["run_d2"](ins).items()}
        except Exception as e:                                        # noqa: BLE001 - report and go on with the other ops
            failures.append(f"{name}: detectron2 call failed: {type(e).__name__}: {e}")
            continue
        np.savez_compressed(os.path.join(args.out, f"d2_{name}.npz"), **{"in_" + k: v.numpy() for k, v in ins.items()},
                            **{"out_" + k: v.numpy() for k, v in ref.items()}, d2_version=np.array(str(getattr(detectron2, "__version__", "?"))))
        got = {k: v.detach().cpu() for k, v in p["run_oracle"](ins).items()}
        bad = compare(name, {k: v.numpy() for k, v in got.items()}, {k: v.numpy() for k, v in ref.items()}, p["tol"])
        print(f"  {name:28s} {'OK' if not bad else 'MISMATCH'}  (tests/golden/d2_{name}.npz written)")
        failures += bad
    for f in failures:
        print("  !! " + f)
    print("the oracle reproduces detectron2 on every pinned op" if not failures else
          f"{len(failures)} mismatches: fix oracle/d2_ops.c / oracle/d2ops.py (and the HIP kernels that follow them), then re-run")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())

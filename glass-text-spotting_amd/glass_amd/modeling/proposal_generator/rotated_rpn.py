"""Rotated RPN (inference) on HIP kernels.

Mirrors reference glass/modeling/proposal_generator/rotated_rpn.py:16-17 (`RotatedRPN(RRPN)`,
which only overrides the training loss) = detectron2 v0.6 RRPN inference [d2-recall]:
StandardRPNHead on p2..p6, RotatedAnchorGenerator, Box2BoxTransformRotated, per-level
top-k, clip, rotated NMS, top `POST_NMS_TOPK_TEST`.  Called at reference
glass/modeling/meta_arch/glass_rcnn.py:87 as `proposal_generator(images, features, None)`.

MI355X-first choices: the two 1x1 heads are one 72-channel conv; anchors are never
materialised (generated from the flat index inside the top-k kernel, only for the 1000
winners per level); the whole selection runs on device with no host sync
(`forward_batched`); the list-of-Instances surface of the reference is a thin wrapper.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

from ...utils.module import InferenceModule

from ...checkpoint import dev, fold_conv
from ...ops import native as K
from ...structures.core import ImageList, Instances, RotatedBoxes
from ...utils.registry import PROPOSAL_GENERATOR_REGISTRY
from ..backbone.resnet_fpn import as_nhwc


def rotated_cell_anchors(size: float, aspect_ratios, angles) -> torch.Tensor:
    """RotatedAnchorGenerator.generate_cell_anchors for one size: ratio-major, angle-minor."""
    rows = []
    area = float(size) ** 2.0
    for r in aspect_ratios:
        w = math.sqrt(area / r)
        h = r * w
        rows.extend([0.0, 0.0, w, h, float(a)] for a in angles)
    return torch.tensor(rows, dtype=torch.float32)


@PROPOSAL_GENERATOR_REGISTRY.register()
class RotatedRPN(InferenceModule):
    def __init__(self, cfg, input_shape=None):
        super().__init__()
        self.in_features = list(cfg.MODEL.RPN.IN_FEATURES)
        ag = cfg.MODEL.ANCHOR_GENERATOR
        assert ag.NAME == "RotatedAnchorGenerator", ag.NAME
        sizes = ag.SIZES if len(ag.SIZES) == len(self.in_features) else list(ag.SIZES) * len(self.in_features)
        self.cell_anchors_cpu = []
        for s in sizes:
            assert len(s) == 1, "one anchor size per level (reference configs)"
            self.cell_anchors_cpu.append(rotated_cell_anchors(s[0], ag.ASPECT_RATIOS[0], ag.ANGLES[0]))
        self.num_anchors = self.cell_anchors_cpu[0].shape[0]
        self.anchor_offset = float(ag.OFFSET)
        self.weights = tuple(float(v) for v in cfg.MODEL.RPN.BBOX_REG_WEIGHTS)
        assert len(self.weights) == 5
        self.pre_nms_topk = int(cfg.MODEL.RPN.PRE_NMS_TOPK_TEST)
        self.post_nms_topk = int(cfg.MODEL.RPN.POST_NMS_TOPK_TEST)
        self.nms_thresh = float(cfg.MODEL.RPN.NMS_THRESH)
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        assert self.min_box_size == 0.0, "MIN_SIZE != 0 is not built"
        self.w: Dict[str, tuple] = {}
        self.cell_anchors: List[torch.Tensor] = []

    def import_weights(self, sd, device, prefix: str = "proposal_generator.") -> None:
        p = prefix + "rpn_head."
        self.w["conv"] = fold_conv(sd, p + "conv", None, device)
        # objectness (A) and anchor_deltas (5A) 1x1 heads merged into one conv
        wl, bl = sd[p + "objectness_logits.weight"], sd[p + "objectness_logits.bias"]
        wd, bd = sd[p + "anchor_deltas.weight"], sd[p + "anchor_deltas.bias"]
        merged = {"m.weight": torch.cat([wl, wd], 0), "m.bias": torch.cat([bl, bd], 0)}
        self.w["heads"] = fold_conv(merged, "m", None, device)
        self.cell_anchors = [dev(c, device) for c in self.cell_anchors_cpu]

    # ------------------------------------------------------------------ device-only path
    def forward_batched(self, feats_nhwc: List[torch.Tensor], image_hw_dev: torch.Tensor):
        """feats_nhwc: level tensors [N,H,W,256]; image_hw_dev: int32 [N,2] (h,w) on device.
        Returns (boxes [N,P,5], logits [N,P], counts int32 [N]) on device, P = POST_NMS_TOPK."""
        A = self.num_anchors
        N = feats_nhwc[0].shape[0]
        device = feats_nhwc[0].device
        ks = [min(self.pre_nms_topk, f.shape[1] * f.shape[2] * A) for f in feats_nhwc]
        S = sum(ks)
        cand_boxes = torch.empty((N, S, 5), dtype=torch.float32, device=device)
        cand_scores = torch.empty((N, S), dtype=torch.float32, device=device)
        cand_level = torch.empty((N, S), dtype=torch.int32, device=device)
        off = 0
        levels, keep_alive = [], []
        for lvl, f in enumerate(feats_nhwc):
            t = K.conv2d_nhwc(f, *self.w["conv"], padding=1, relu=1)
            head = K.conv2d_nhwc(t, *self.w["heads"], out_dtype=torch.float32)   # [N,H,W,6A]: logits | deltas (fp32: top-k input)
            keep_alive.append(head)
            ld = head.shape[-1]
            levels.append({"logits": head, "deltas": head.view(-1)[A:], "ldl": ld, "ldd": ld, "H": f.shape[1],
                           "W": f.shape[2], "stride": 2 ** (lvl + 2), "cell_anchors": self.cell_anchors[lvl],
                           "topk": self.pre_nms_topk, "slot_off": off})
            off += ks[lvl]
        # one chip-wide radix select over all levels and images (6 launches per step)
        K.rpn_topk_decode(levels, N, A, self.anchor_offset, self.weights, cand_boxes, cand_scores, cand_level)
        boxes, scores, _, counts = K.rotated_nms_select(
            cand_boxes, cand_scores, cand_level, None, image_hw_dev, float("-inf"), self.nms_thresh, self.post_nms_topk,
            K.NMS_CLIP | K.NMS_DROP_EMPTY)
        return boxes, scores, counts

    # ------------------------------------------------------------------ reference surface
    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], gt_instances=None):
        assert gt_instances is None and not self.training, "inference only"
        feats = [as_nhwc(features[f]) for f in self.in_features]
        hw = K.upload(images.image_sizes, torch.int32, feats[0].device)
        boxes, scores, counts = self.forward_batched(feats, hw)
        cnt = counts.cpu().tolist()
        out = []
        for n, image_size in enumerate(images.image_sizes):
            inst = Instances(image_size)
            inst.proposal_boxes = RotatedBoxes(boxes[n, : cnt[n]])
            inst.objectness_logits = scores[n, : cnt[n]]
            out.append(inst)
        return out, {}

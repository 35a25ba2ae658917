"""Rotated Fast R-CNN box predictor + inference on HIP kernels.

Mirrors reference glass/modeling/roi_heads/rotated_fast_rcnn.py: `RotatedFastRCNNOutputLayers`
(:494-638: cls_score 2048->K+1, bbox_pred ->5, orientation_pred ->4), `RotatedFastRCNNOutputs.
{predict_boxes :335-342, predict_probs :480-482, predict_orientations :484-491, inference
:344-373}` and `fast_rcnn_inference_single_image_rotated` (:88-148).

The three linears are one 11-row MFMA GEMM; softmax / delta decode run in one kernel; the
finite filter, clip, score threshold, rotated NMS and top-k run in one kernel per batch with
per-image semantics identical to running the reference image by image.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from ...utils.module import InferenceModule

from ...checkpoint import conv_weight, dev
from ...ops import native as K
from ...structures.core import Instances, RotatedBoxes


class RotatedFastRCNNOutputLayers(InferenceModule):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        assert self.num_classes == 1, "one foreground class ('word') is built (all reference configs)"
        self.orientation_on = bool(cfg.MODEL.ORIENTATION_ON)     # off in glass_finetune_textocr.yaml
        self.class_names = [n.lower() for n in cfg.MODEL.ROI_HEADS.CLASS_NAMES]
        self.weights = tuple(float(v) for v in cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
        self.test_score_thresh = float(cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST)
        self.test_nms_thresh = float(cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST)
        self.test_topk_per_image = int(cfg.TEST.DETECTIONS_PER_IMAGE)
        self.w = {}

    def import_weights(self, sd, device, prefix: str) -> None:
        names = ("cls_score", "bbox_pred") + (("orientation_pred",) if self.orientation_on else ())
        self.w = {"w": conv_weight(torch.cat([sd[prefix + n + ".weight"] for n in names], 0), device),
                  "b": dev(torch.cat([sd[prefix + n + ".bias"] for n in names], 0), device)}

    def forward(self, x: torch.Tensor):
        """x [R,2048] -> (scores [R,2], deltas [R,5], orientation logits [R,4])."""
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        y = K.linear(x.contiguous(), self.w["w"], self.w["b"])
        orient = y[:, 7:11].contiguous() if self.orientation_on else None
        return y[:, 0:2].contiguous(), y[:, 2:7].contiguous(), orient

    def inference_batched(self, predictions, proposal_boxes: torch.Tensor, proposal_counts: torch.Tensor,
                          image_hw_dev: torch.Tensor):
        """predictions for N*P padded proposal slots; proposal_boxes [N,P,5]; counts int32 [N].
        Returns (boxes [N,K,5], scores [N,K], kept slot index [N,K], orientations [N*P,2], counts [N])."""
        scores, deltas, orient = predictions
        N, P, _ = proposal_boxes.shape
        if orient is None:          # orientation head off: decode with dummy logits, field dropped later
            orient = torch.zeros((scores.shape[0], 4), dtype=torch.float32, device=scores.device)
        boxes, fg, orient2 = K.box_decode(scores, deltas, orient, proposal_boxes.view(-1, 5), self.weights)
        ob, os_, oi, oc = K.rotated_nms_select(boxes.view(N, P, 5), fg.view(N, P), None, proposal_counts, image_hw_dev,
                                               self.test_score_thresh, self.test_nms_thresh,
                                               self.test_topk_per_image, K.NMS_CLIP)
        return ob, os_, oi, (orient2.view(N, P, 2) if self.orientation_on else None), oc

    def inference(self, predictions, proposals: List[Instances]):
        """reference surface: list[Instances] with proposal_boxes -> (list[Instances], kept indices)."""
        device = predictions[0].device
        counts = [len(p) for p in proposals]
        N, P = len(proposals), max(counts + [1])
        pb = torch.zeros((N, P, 5), dtype=torch.float32, device=device)
        predictions = tuple(p for p in predictions if p is not None)
        sc = [torch.zeros((N * P, p.shape[1]), dtype=torch.float32, device=device) for p in predictions]
        start = 0
        for n, p in enumerate(proposals):
            pb[n, : counts[n]] = p.proposal_boxes.tensor
            for dst, src in zip(sc, predictions):
                dst[n * P: n * P + counts[n]] = src[start: start + counts[n]]
            start += counts[n]
        hw = K.upload([p.image_size for p in proposals], torch.int32, device)
        cnt = K.upload(counts, torch.int32, device)
        ob, os_, oi, orient2, oc = self.inference_batched(tuple(sc) + ((None,) if len(sc) == 2 else ()), pb, cnt, hw)
        return self.to_instances(ob, os_, oi, orient2, oc, [p.image_size for p in proposals])

    @staticmethod
    def to_instances(ob, os_, oi, orient2, oc, image_sizes) -> Tuple[List[Instances], List[torch.Tensor]]:
        cnt = oc.cpu().tolist()
        results, kept = [], []
        for n, image_size in enumerate(image_sizes):
            k = cnt[n]
            idx = oi[n, :k].long()
            r = Instances(image_size)
            r.pred_boxes = RotatedBoxes(ob[n, :k])
            r.scores = os_[n, :k]
            r.pred_classes = torch.zeros((k,), dtype=torch.int64, device=ob.device)
            if orient2 is not None:
                r.orientations = orient2[n][idx]
            results.append(r)
            kept.append(idx)
        return results, kept

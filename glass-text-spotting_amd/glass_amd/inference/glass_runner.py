"""`GlassRunner`: the single-image predictor the reference ships, on the HIP pipeline.

Mirrors reference glass/inference/glass_runner.py:20-153: cfg set-up (:31-39), model build and
checkpoint load (:52-60), channel handling (:83-87), on-device bilinear resize policy
(`get_inference_scale_ratio` :111-121, `_image_to_tensor` :123-148), model call (:93-96),
un-scaling of boxes (:100-102) and post-processing (:106).
`run_batch` is the throughput form: many images per step, identical per-image results.
"""
from __future__ import annotations

import logging
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..config import get_glass_cfg
from ..modeling.meta_arch.glass_rcnn import build_model
from ..modeling.recognition.text_encoder import TextEncoder
from ..ops import native as K
from ..postprocess import build_post_processor
from ..structures.core import Instances
from ..utils.host import limit_host_threads


class GlassRunner:
    def __init__(self, model_path: Optional[str], config_path: Optional[str], opts: List[str] = None, post_process=True,
                 cfg=None, state_dict=None):
        self.logger = logging.getLogger(__name__)
        limit_host_threads()                                  # utils/host.py: the host side is launch glue
        self.cfg = (cfg if cfg is not None else get_glass_cfg(config_path, opts)).clone()
        self.model_path, self.config_path, self.post_process_flag = model_path, config_path, post_process
        self.model = build_model(self.cfg)
        self.model.eval()
        self.device = self.model.device
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        elif model_path:
            self.model.load_checkpoint(model_path)
        self.min_target_size = self.cfg.INPUT.MIN_SIZE_TEST
        self.max_target_size = self.cfg.INPUT.MAX_SIZE_TEST
        self.max_upscale_ratio = self.cfg.INPUT.MAX_UPSCALE_RATIO
        self.input_format = self.cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR", "GREY"], self.input_format
        self.text_encoder = TextEncoder(self.cfg)
        self.post_processor = build_post_processor(self.cfg)

    def get_inference_scale_ratio(self, image_shape):
        height, width = image_shape[:2]
        m = max(height, width)
        if m > self.max_target_size:
            return self.max_target_size / m
        if m < self.min_target_size:
            return min(self.max_upscale_ratio, self.min_target_size / m)
        return 1

    def _image_to_tensor(self, original_image: np.ndarray):
        """uint8 HWC on host -> float CHW on device, resized by the reference policy; one H2D copy
        of the uint8 image and one fused convert+resize kernel."""
        if self.input_format == "GREY":
            # reference glass/utils/common_utils.py:29-43 (rgb2grey, three_channels=True), on the host uint8 image as
            # the reference does: Y'709 weights on channels 0,1,2, truncated to uint8, replicated three times
            g = np.uint8(0.2125 * original_image[:, :, 0] + 0.7154 * original_image[:, :, 1] + 0.0721 * original_image[:, :, 2])
            original_image = np.repeat(g[:, :, None], 3, axis=2)
        height, width = original_image.shape[:2]
        scale_ratio = self.get_inference_scale_ratio(original_image.shape)
        if scale_ratio != 1:
            nh, nw = int(np.round(scale_ratio * height)), int(np.round(scale_ratio * width))
        else:
            nh, nw = height, width
        u8 = torch.from_numpy(np.ascontiguousarray(original_image)).to(self.device)
        chw = K.image_u8hwc_to_chw(u8, (nh, nw), flip_channels=(self.input_format == "RGB"))
        return chw, scale_ratio

    def run_batch(self, images: Sequence[np.ndarray]) -> List[Instances]:
        """Many images per step with per-image results identical to image-by-image calls.  detectron2
        pads a batch to its largest image and the pad region is visible to the backbone/RPN, so images
        are grouped by their own padded shape (multiple of 32) and each group is one model call."""
        inputs, ratios, shapes = [], [], []
        for im in images:
            t, r = self._image_to_tensor(im)
            inputs.append({"image": t, "height": t.shape[1], "width": t.shape[2]})
            ratios.append(r)
            shapes.append(im.shape[:2])
        d = self.model.backbone.size_divisibility
        groups = {}
        for i, inp in enumerate(inputs):
            key = ((inp["height"] + d - 1) // d * d, (inp["width"] + d - 1) // d * d)
            groups.setdefault(key, []).append(i)
        out = [None] * len(inputs)
        for idxs in groups.values():
            raw = self.model([inputs[i] for i in idxs])
            det = getattr(raw, "batch", None)          # this call's padded device-resident batch (pipeline.StepOutput)
            if self.post_process_flag and det is not None:
                # un-scale + the whole word post-processing for the group in one kernel (no per-image host loop)
                if det.text is None and sum(det.counts_host) > 0:
                    raise RuntimeError("recognizer output missing")
                sc = torch.tensor([[1.0 / ratios[i], 1.0 / ratios[i]] for i in idxs], dtype=torch.float32).to(self.device)
                text = det.text
                if text is None:          # no detections in the whole group
                    T = self.cfg.MODEL.ROI_RECOGNIZER_HEAD.MAX_WORD_LENGTH + 1
                    C = len(self.cfg.MODEL.ROI_RECOGNIZER_HEAD.CHARACTER_SET) + 2
                    text = torch.zeros(tuple(det.scores.shape) + (T, C), dtype=torch.float32, device=self.device)
                res = self.post_processor.process_padded(det.boxes, det.scores, det.counts_dev, text, sc,
                                                         [shapes[i] for i in idxs], {"orientations": det.orient})
                for i, r in zip(idxs, res):
                    out[i] = r
            else:
                for i, res in zip(idxs, raw):
                    preds = res["instances"]
                    if ratios[i] != 1:
                        preds.pred_boxes.scale(1 / ratios[i], 1 / ratios[i])
                    preds._image_size = tuple(shapes[i])
                    out[i] = preds
        for r in out:
            self.logger.info(f"Post-processing output is {len(r)} word instances")
        return out

    def __call__(self, original_image: np.ndarray) -> Instances:
        return self.run_batch([original_image])[0]

    def preds_boxes_to_polygons(self, pred_boxes):
        return self.post_processor.boxes_to_polygons(boxes=pred_boxes.tensor)

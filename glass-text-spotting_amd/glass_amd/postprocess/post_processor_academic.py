"""`PostProcessorAcademic` + `detector_postprocess` + `get_instances_text`.

Behavioural mirror of reference glass/postprocess/post_processor_academic.py:19-35 (text-score
filter after the rotated-box post-process), :118-178 (`detector_postprocess`: scale, clip, drop
empty; the mask paste branch is out of scope) and glass/evaluation/text_evaluator.py:323-348
(`get_instances_text`: argmax -> TextEncoder.decode_prod_v2 -> strip one leading/trailing
special character).
"""
from __future__ import annotations

import torch

from ..modeling.recognition.text_encoder import TextEncoder
from ..structures.core import Instances
from .post_processor_rotated_boxes import POST_PROCESSOR_REGISTRY, PostProcessorRotatedBoxes

_SPECIAL = str("'!?.:,*+\"()·[]/\\#$%;<=>@^_`{|}~")


def strip_special(text: str) -> str:
    """strip ONE leading and ONE trailing special character (text_evaluator.py:337-343)"""
    if len(text) > 0 and _SPECIAL.find(text[0]) > -1:
        text = text[1:]
    if len(text) > 0 and _SPECIAL.find(text[-1]) > -1:
        text = text[:-1]
    return text


def get_instances_text(text_probs, text_encoder, onlyRemoveFirstLastCharacter=True):
    if len(text_probs):
        text_probs = text_probs.detach().cpu()
        pred_probs, pred_idx = text_probs.max(dim=2)
        text_probs = text_probs.numpy()
        objs = text_encoder.decode_prod_v2(pred_probs=pred_probs.numpy(), pred_indices=pred_idx.numpy())
        pred_text = [o["text"] for o in objs]
        pred_scores = [o["score"] for o in objs]
        if onlyRemoveFirstLastCharacter:
            for i in range(len(pred_text)):
                if len(pred_text[i]) > 0 and _SPECIAL.find(pred_text[i][0]) > -1:
                    pred_text[i] = pred_text[i][1:]
                if len(pred_text[i]) > 0 and _SPECIAL.find(pred_text[i][-1]) > -1:
                    pred_text[i] = pred_text[i][:-1]
    else:
        pred_text, pred_scores, text_probs = [], [], []
    return pred_text, pred_scores, text_probs


@POST_PROCESSOR_REGISTRY.register()
class PostProcessorAcademic(PostProcessorRotatedBoxes):
    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self.text_threshold = cfg.POST_PROCESSING.TEXT_THRESHOLD
        self.text_encoder = TextEncoder(cfg)

    @staticmethod
    def resize_boxes(preds: Instances, ratio: float, axis="both"):
        """Inflate boxes by `ratio` of their own width / height, then clip (reference :36-62; used by
        GlassRCNN._postprocess when POST_PROCESSING.INFLATE_RATIO is set)."""
        if len(preds) == 0:
            return preds
        boxes = preds.pred_boxes.tensor
        if axis == "both":
            delta_x, delta_y = ratio * boxes[:, 2], ratio * boxes[:, 3]
        elif axis == "vertical":
            delta_x, delta_y = 0, ratio * boxes[:, 3]
        elif axis == "horizontal":
            delta_x, delta_y = ratio * boxes[:, 2], 0
        else:
            raise Exception('Please provide an axis value of either "both"/"horizontal"/"vertical')
        boxes[:, 2] += delta_x
        boxes[:, 3] += delta_y
        preds.pred_boxes.tensor = boxes
        preds.pred_boxes.clip(preds.image_size)
        return preds

    @staticmethod
    def drop_overlapping_boxes(preds: Instances, ioa_threshold: float, valid_score: float, minimal_ioa_thresh=0.01):
        """Replace both boxes of every strongly overlapping confident pair by the larger one, then NMS at 0.99 drops
        the duplicate with the lower score (reference :64-116; POST_PROCESSING.DROP_OVERLAPPING).  IoA and NMS run on
        the HIP kernels (structures/boxes.py); the pair bookkeeping is a few index ops on <= 100 boxes.  As written the
        reference hands RotatedBoxes objects to its tensor-only pairwise_ioa_rotated (AttributeError on `.shape`); this
        is the evidently intended computation on the box tensors."""
        from ..structures.boxes import nms_rotated, pairwise_ioa_rotated
        if len(preds) == 0:
            return preds
        ioa = pairwise_ioa_rotated(preds.pred_boxes.tensor, preds.pred_boxes.tensor)
        boxes = preds.pred_boxes.tensor
        scores = preds.scores
        assert boxes.shape[1] == 5
        areas = boxes[:, 2] * boxes[:, 3]
        pairs = torch.nonzero(ioa.fill_diagonal_(0).triu() >= minimal_ioa_thresh)
        if len(pairs) == 0:
            return preds
        min_pair_score = torch.min(scores[pairs[:, 0]], scores[pairs[:, 1]])
        combined = (min_pair_score >= valid_score) & (ioa[pairs[:, 0], pairs[:, 1]] >= ioa_threshold)
        if (~combined).all():
            return preds
        op = pairs[combined]
        larger = torch.where((areas[op[:, 0]] > areas[op[:, 1]])[..., None], boxes[op[:, 0]], boxes[op[:, 1]])
        # duplicate indices: the reference's index_put keeps the LAST write on CPU; done in order here so the
        # device result is the same deterministic one
        hb, hl = boxes.cpu(), larger.cpu()
        for col in (0, 1):
            for k, i in enumerate(op[:, col].tolist()):
                hb[i] = hl[k]
        preds.pred_boxes.tensor = hb.to(boxes.device)
        keep = nms_rotated(preds.pred_boxes.tensor, preds.scores, iou_threshold=0.99)
        return preds[keep]

    def host_call(self, preds, scale_ratio=1, **kwargs):
        preds = super().host_call(preds)
        texts, text_scores, _ = get_instances_text(preds.pred_text_prob, self.text_encoder)
        keep = torch.tensor(text_scores) >= self.text_threshold
        return preds[keep.to(preds.pred_boxes.device) if len(text_scores) else torch.zeros((0,), dtype=torch.bool)]


def detector_postprocess(results: Instances, output_height, output_width, mask_threshold=0.5) -> Instances:
    ow = output_width.float() if isinstance(output_width, torch.Tensor) else output_width
    oh = output_height.float() if isinstance(output_height, torch.Tensor) else output_height
    scale_x, scale_y = ow / results.image_size[1], oh / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    if results.has("pred_boxes"):
        output_boxes = results.pred_boxes
    elif results.has("proposal_boxes"):
        output_boxes = results.proposal_boxes
    else:
        return results
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results = results[output_boxes.nonempty()]
    if results.has("pred_masks"):          # reference :167-173: paste_masks_in_image on the scaled rotated boxes
        from ..ops import native as K
        results.pred_masks = K.paste_rotated_masks(results.pred_masks[:, 0, :, :].contiguous(),
                                                   results.pred_boxes.tensor.contiguous(), results.image_size,
                                                   threshold=mask_threshold)
    if results.has("pred_rboxes"):         # reference :174-176.  forward_with_given_boxes aliases pred_rboxes to
        # pred_boxes (recognizers_hybrid_head.py:599): unless an indexing step in between (filter_small_boxes) broke
        # the alias, the in-place scale above already went through pred_rboxes and this one scales it a second time.
        results.pred_rboxes.scale(scale_x, scale_y)
        results.pred_rboxes.clip(results.image_size)
    return results

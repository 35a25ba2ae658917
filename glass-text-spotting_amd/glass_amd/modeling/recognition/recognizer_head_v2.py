"""Recognizer head (`RecognizerRCNNHeadV3`): CNN -> BiLSTM encoder -> attention decoder.

Mirrors reference glass/modeling/recognition/recognizer_head_v2.py:291-345 and the eval branch
of `BaseRecognizerRCNNHead.forward` (:150-163): empty input returns the instances untouched,
otherwise `pred_text_prob` [Ri,26,97] is attached per image.
"""
from __future__ import annotations

from typing import List

import torch

from ...ops import native as K
from ...utils.module import InferenceModule

from ...structures.core import Instances
from ...utils.registry import Registry
from .recognizer_backbone import build_recognizer_backbonev2
from .recognizer_decoder import build_recognizer_decoderv2
from .recognizer_encoder import build_recognizer_encoderv2
from .text_encoder import TextEncoder

ROI_RECOGNIZER_HEAD_REGISTRY = Registry("ROI_RECOGNIZER_HEAD")


@ROI_RECOGNIZER_HEAD_REGISTRY.register()
class RecognizerRCNNHeadV3(InferenceModule):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.backbone = build_recognizer_backbonev2(cfg, input_shape)
        self.encoder = build_recognizer_encoderv2(cfg, input_shape)
        self.decoder = build_recognizer_decoderv2(cfg, input_shape)
        self.max_word_length = cfg.MODEL.ROI_RECOGNIZER_HEAD.MAX_WORD_LENGTH
        self.class_ind = cfg.MODEL.ROI_RECOGNIZER_HEAD.CLASS_IND
        self.text_encoder = TextEncoder(cfg)

    def import_weights(self, sd, device, prefix: str) -> None:
        self.backbone.import_weights(sd, device, prefix + "backbone.")
        self.encoder.import_weights(sd, device, prefix + "encoder.")
        self.decoder.import_weights(sd, device, prefix + "decoder.")

    rnn_override = None       # "steps" once K.HandoffGuard.STICKY_AFTER hand-off give-ups were seen by this head
    rnn_giveups = 0

    def forward_nhwc(self, x: torch.Tensor, roi_image: torch.Tensor, num_images: int, guard=None) -> torch.Tensor:
        """x [R,8,32,256] fused features -> probabilities [R,26,97].
        `guard` (ops.native.HandoffGuard): the one-launch BiLSTM / decoder kernels report a hand-off that gave up into
        `guard.status`, and `guard.retry` re-runs encoder + decoder of THIS call on the step kernels - the caller resolves the
        guard at its next host read-back (GeneralizedRCNN._postprocess_batched_g: the surviving-count read).  Without a guard
        the status is read here (one small synchronising copy): no path returns text from a dead hand-off."""
        f = self.backbone.forward_nhwc(x)
        own = guard is None
        if own:
            guard = K.HandoffGuard(x.device, owner=self)
        elif guard.owner is None:
            guard.owner = self

        def run(rnn, status):
            enc = self.encoder.forward_nhwc(f, rnn=rnn, status=status)
            return self.decoder(enc, roi_image=roi_image, num_images=num_images, rnn=rnn, status=status)
        if self.rnn_override == "steps":
            return run("steps", None)
        out = run(None, guard.status)
        guard.retry = lambda: run("steps", None)
        if own:
            again = guard.resolve(guard.status.item())
            return out if again is None else again
        return out

    def forward(self, x: torch.Tensor, instances: List[Instances]):
        assert not self.training, "inference only"
        if x.shape[0] == 0:
            return instances
        from ..backbone.resnet_fpn import as_nhwc
        counts = [len(i) for i in instances]
        roi_image = K.upload(torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts)),
                             torch.int32, x.device)
        preds = self.forward_nhwc(as_nhwc(x), roi_image, len(counts))
        for p, inst in zip(preds.split(counts, dim=0), instances):
            inst.pred_text_prob = p
        return instances


def build_recognizer_head(cfg, input_shape):
    return ROI_RECOGNIZER_HEAD_REGISTRY.get(cfg.MODEL.ROI_RECOGNIZER_HEAD.NAME)(cfg, input_shape)

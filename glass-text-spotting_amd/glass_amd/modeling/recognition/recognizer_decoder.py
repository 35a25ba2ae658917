"""Attention decoder (`ASTER_V2` -> `AttentionRecognitionHead.sample`) as one persistent HIP kernel.

Mirrors reference glass/modeling/recognition/recognizer_decoder.py:65-93 and
prediction_aster.py:63-99 (greedy sampling, eos index 0, pre-zeroed [R,26,97] output with
the batch-global early break), AttentionUnit :247-266, DecoderUnit :291-302.
xEmbed(x) is recomputed at every step by the reference; it is step-invariant, so it is one
MFMA GEMM here.
"""
from __future__ import annotations

from typing import Optional

import torch

from ...utils.module import InferenceModule

from ...checkpoint import conv_weight, dev
from ...ops import native as K
from ...utils.registry import Registry

RECOGNIZER_DECODER_REGISTRY = Registry("RECOGNIZER_DECODER")


def build_recognizer_decoderv2(cfg, input_shape):
    return RECOGNIZER_DECODER_REGISTRY.get(cfg.MODEL.ROI_RECOGNIZER_HEAD.RECOGNIZER_HEAD.DECODER.NAME)(cfg, input_shape)


@RECOGNIZER_DECODER_REGISTRY.register()
class ASTER_V2(InferenceModule):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.num_classes = len(cfg.MODEL.ROI_RECOGNIZER_HEAD.CHARACTER_SET) + 2
        self.max_word_len = int(cfg.MODEL.ROI_RECOGNIZER_HEAD.MAX_WORD_LENGTH) + 1
        self.in_channels = input_shape.channels
        self.w = {}

    def import_weights(self, sd, device, prefix: str) -> None:
        q = prefix + "recognizer.decoder."
        f = lambda k: sd[q + k].float()
        self.w = {
            "sW": dev(K.pack_kblocked(f("attention_unit.sEmbed.weight")), device),
            "sB": dev(f("attention_unit.sEmbed.bias"), device),
            "xW": conv_weight(f("attention_unit.xEmbed.weight"), device),
            "xB": dev(f("attention_unit.xEmbed.bias"), device),
            "wW": dev(f("attention_unit.wEmbed.weight").reshape(-1), device),
            "wB": dev(f("attention_unit.wEmbed.bias").reshape(-1), device),
            "emb": dev(f("tgt_embedding.weight"), device),
            "w_ih": dev(f("gru.weight_ih_l0"), device),          # [3D, 2D] row-major (MFMA GRU kernel)
            "w_hh": dev(f("gru.weight_hh_l0"), device),          # [3D, D]
            "b_ih": dev(f("gru.bias_ih_l0"), device),
            "b_hh": dev(f("gru.bias_hh_l0"), device),
            "fcW": dev(K.pack_kblocked(f("fc.weight")), device),
            "fcB": dev(f("fc.bias"), device),
            "temperature": float(sd[q + "temperature"].reshape(-1)[0]) if (q + "temperature") in sd else 1.0,
        }

    def beam_search(self, x, beam_width, eos):
        """reference prediction_aster.py:101-222: never called on the inference path (`forward` -> `sample`, :63-99,
        greedy) nor by any tool of the reference; not built."""
        raise NotImplementedError("AttentionRecognitionHead.beam_search (reference prediction_aster.py:101-222) is dead code in the "
                                  "reference's inference path (greedy `sample` is what runs); this build implements `sample` only")

    def forward(self, features: torch.Tensor, labels=None, roi_image: Optional[torch.Tensor] = None,
                num_images: int = 1) -> torch.Tensor:
        """features [R,T,D] -> probabilities [R,max_word_len,num_classes].  `roi_image` (int32 [R],
        image id per RoI) scopes the reference's early break to one image's RoIs; default: one call =
        one image, as in the reference."""
        assert labels is None and not self.training, "inference only"
        x = features.contiguous()
        R, T, D = x.shape
        if roi_image is None:
            roi_image = torch.zeros((R,), dtype=torch.int32, device=x.device)
            num_images = 1
        xproj = K.linear(x.view(R * T, D), self.w["xW"], self.w["xB"]).view(R, T, D)
        return K.attention_decode(x, xproj, self.w, roi_image, num_images, self.num_classes, self.max_word_len, 0)

"""Attention decoder (`ASTER_V2` -> `AttentionRecognitionHead.sample`) behind ONE ABI call: glass_attention_decode_persistent (ONE
launch for all steps: GRU rows and sEmbed resident in registers, h / context handed between the 16 workgroups of a 16-RoI group
inside the launch) or, with Routing.rnn == "steps", glass_attention_decode (two kernels per decoding step spread over the chip);
arg-max feedback on the device, the per-image early break applied by a mask kernel afterwards.

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
        # the one-launch decoder (glass_attention_decode_persistent): sEmbed as torch stores it, and the embedding half of the
        # GRU input as a table - W_ih[:, :D] emb[c] + b_ih for every class c, computed once (fp64, stored fp32)
        D = f("gru.weight_hh_l0").shape[1]
        self.w["sW_rm"] = dev(f("attention_unit.sEmbed.weight").contiguous(), device)
        self.w["emb_gi"] = dev((f("tgt_embedding.weight").double() @ f("gru.weight_ih_l0")[:, :D].double().t()
                                + f("gru.bias_ih_l0").double()).float().contiguous(), device)

    def beam_search(self, x: torch.Tensor, beam_width: int, eos: int = 0):
        """`AttentionRecognitionHead.beam_search` (reference prediction_aster.py:101-222; never called by the reference's
        inference path, which samples greedily): x [B,T,D] -> (symbols [B, max_word_len] of the best beam, its score [B]).
        The per-step arithmetic - attention, GRU cell, fc - is one `glass_attention_decode_step` on the B x beam_width
        inflated batch; the search itself is the reference's host logic on small device tensors: log-softmax + running
        scores, top-k over beam x classes, predecessor re-indexing of the state, <eos> beams frozen at -inf, and the
        back-tracking pass including its "ended sequences replace the worst beams" bookkeeping.  One deliberate difference:
        `candidates / num_classes` is floor division here, as it was under the torch version the reference was written for -
        on a current torch the reference line yields a float index and `index_select` raises (oracle/make_golden.py --beam
        generates the golden from the reference function with exactly that operator restored)."""
        x = x.contiguous()
        B, T, D = x.shape
        k, C, L = int(beam_width), self.num_classes, self.max_word_len
        dev_ = x.device
        xproj = K.linear(x.view(B * T, D), self.w["xW"], self.w["xB"]).view(B, T, D)
        xi = x.unsqueeze(1).expand(B, k, T, D).reshape(B * k, T, D).contiguous()              # ABC -> AABBCC
        xpi = xproj.unsqueeze(1).expand(B, k, T, D).reshape(B * k, T, D).contiguous()
        state = torch.zeros((B * k, D), dtype=torch.float32, device=dev_)
        pos_index = (torch.arange(B, device=dev_) * k).view(-1, 1)
        sequence_scores = torch.full((B * k, 1), -float("inf"), device=dev_)
        sequence_scores[torch.arange(B, device=dev_) * k] = 0.0
        y_prev = torch.zeros((B * k,), dtype=torch.int32, device=dev_)
        stored_scores, stored_predecessors, stored_emitted_symbols = [], [], []
        for _ in range(L):
            logits, _, state = K.attention_decode_step(xi, xpi, self.w, state, y_prev, C)
            log_softmax_output = torch.log_softmax(logits, dim=1)
            sequence_scores = sequence_scores.repeat(1, C) + log_softmax_output
            scores, candidates = sequence_scores.view(B, -1).topk(k, dim=1)
            y_prev = (candidates % C).view(B * k).to(torch.int32)
            sequence_scores = scores.view(B * k, 1)
            predecessors = (candidates // C + pos_index.expand_as(candidates)).view(B * k, 1)
            state = state.index_select(0, predecessors.squeeze(1)).contiguous()
            stored_scores.append(sequence_scores.clone())
            sequence_scores = sequence_scores.masked_fill(y_prev.view(-1, 1) == eos, -float("inf"))
            stored_predecessors.append(predecessors)
            stored_emitted_symbols.append(y_prev.long())
        # ---- back-tracking (host bookkeeping of the reference, :165-222), on CPU copies of the small decision tables
        sc = [t.cpu() for t in stored_scores]
        pr = [t.cpu() for t in stored_predecessors]
        sy = [t.cpu() for t in stored_emitted_symbols]
        pos = pos_index.cpu()
        p = []
        lens = [[L] * k for _ in range(B)]
        sorted_score, sorted_idx = sc[-1].view(B, k).topk(k)
        s = sorted_score.clone()
        batch_eos_found = [0] * B
        t_pred = (sorted_idx + pos.expand_as(sorted_idx)).view(B * k)
        for t in range(L - 1, -1, -1):
            current_symbol = sy[t].index_select(0, t_pred)
            t_pred = pr[t].index_select(0, t_pred).squeeze(1)
            eos_indices = sy[t].eq(eos).nonzero()
            for i in range(eos_indices.size(0) - 1, -1, -1):
                idx = int(eos_indices[i][0])
                b_idx = idx // k
                res_k_idx = k - (batch_eos_found[b_idx] % k) - 1
                batch_eos_found[b_idx] += 1
                res_idx = b_idx * k + res_k_idx
                t_pred[res_idx] = pr[t][idx, 0]
                current_symbol[res_idx] = sy[t][idx]
                s[b_idx, res_k_idx] = sc[t][idx, 0]
                lens[b_idx][res_k_idx] = t + 1
            p.append(current_symbol)
        s, re_sorted_idx = s.topk(k)
        re_sorted_idx = (re_sorted_idx + pos.expand_as(re_sorted_idx)).view(B * k)
        p = [step.index_select(0, re_sorted_idx).view(B, k, -1) for step in reversed(p)]
        p = torch.cat(p, -1)[:, 0, :]
        return p.to(dev_), s[:, 0].to(dev_)

    def forward(self, features: torch.Tensor, labels=None, roi_image: Optional[torch.Tensor] = None,
                num_images: int = 1, rnn=None, status=None) -> torch.Tensor:
        """features [R,T,D] -> probabilities [R,max_word_len,num_classes].  `roi_image` (int32 [R],
        image id per RoI) scopes the reference's early break to one image's RoIs; default: one call =
        one image, as in the reference.  `rnn` / `status`: as in BiLSTMBlockV2.forward_nhwc; without a `status` word
        (the reference's call surface) the hand-off status is read here and a give-up re-runs the decoder on the step kernels."""
        assert labels is None and not self.training, "inference only"
        if status is None and rnn != "steps":
            guard = K.HandoffGuard(features.device, owner=self)
            out = self.forward(features, labels, roi_image, num_images, rnn=getattr(self, "rnn_override", None) or rnn, status=guard.status)
            guard.retry = lambda: self.forward(features, labels, roi_image, num_images, rnn="steps")
            again = guard.resolve(guard.status.item())
            return out if again is None else again
        x = features.contiguous()
        R, T, D = x.shape
        if roi_image is None:
            roi_image = torch.zeros((R,), dtype=torch.int32, device=x.device)
            num_images = 1
        xproj = K.linear(x.view(R * T, D), self.w["xW"], self.w["xB"]).view(R, T, D)
        return K.attention_decode(x, xproj, self.w, roi_image, num_images, self.num_classes, self.max_word_len, 0,
                                  mode=rnn if rnn is not None else K.routing_of(self.w["xW"]).rnn, status=status)

"""Sequence encoder (`BiLSTMBlockV2`): mean over H, 2 x (BiLSTM(256,256) + Linear(512,256)).

Mirrors reference glass/modeling/recognition/recognizer_encoder.py:101-144.  The input
projections of both directions are one MFMA GEMM ([R*T,256] x [256,2048]); the recurrence is one ABI call per
layer: glass_bilstm_recurrence_persistent (ONE launch: W_hh resident in registers, h_t handed between the 8 workgroups of a
16-RoI chain inside the launch) or, with Routing.rnn == "steps", glass_bilstm_recurrence (one chip-wide kernel per time step).
"""
from __future__ import annotations

import torch

from ...utils.module import InferenceModule

from ...checkpoint import conv_weight, dev
from ...ops import native as K
from ...utils.registry import Registry

RECOGNIZER_ENCODER_REGISTRY = Registry("RECOGNIZER_ENCODER")


def build_recognizer_encoderv2(cfg, input_shape):
    return RECOGNIZER_ENCODER_REGISTRY.get(cfg.MODEL.ROI_RECOGNIZER_HEAD.RECOGNIZER_HEAD.ENCODER.NAME)(cfg, input_shape)


@RECOGNIZER_ENCODER_REGISTRY.register()
class BiLSTMBlockV2(InferenceModule):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.hidden = input_shape.channels
        self.num_layers = cfg.MODEL.ROI_RECOGNIZER_HEAD.RECOGNIZER_HEAD.ENCODER.NUM_OF_LAYERS
        self.layers = []

    def import_weights(self, sd, device, prefix: str) -> None:
        self.layers = []
        for layer in range(self.num_layers):
            q = f"{prefix}bilsm_stack.{layer}."
            w_ih = torch.cat([sd[q + "rnn.weight_ih_l0"], sd[q + "rnn.weight_ih_l0_reverse"]], 0)      # [2*4H, I]
            b = torch.cat([sd[q + "rnn.bias_ih_l0"] + sd[q + "rnn.bias_hh_l0"],
                           sd[q + "rnn.bias_ih_l0_reverse"] + sd[q + "rnn.bias_hh_l0_reverse"]], 0)
            w_hh = torch.stack([sd[q + "rnn.weight_hh_l0"].float(), sd[q + "rnn.weight_hh_l0_reverse"].float()], 0)
            self.layers.append({"w_ih": conv_weight(w_ih, device), "b": dev(b, device), "w_hh": dev(w_hh, device),
                                "lin_w": conv_weight(sd[q + "linear.weight"], device), "lin_b": dev(sd[q + "linear.bias"], device)})

    def forward_nhwc(self, feats: torch.Tensor, rnn=None, status=None) -> torch.Tensor:
        """feats [R,H,W,C] -> [R,W,C].  `rnn`: overrides the weights' `Routing.rnn` ("steps": the fall-back of a hand-off that
        gave up); `status`: this call's hand-off status word (ops.native.new_handoff_status) for the persistent kernel."""
        x = K.mean_over_h(feats)
        R, T, _ = x.shape
        for L in self.layers:
            xg = K.linear(x.view(R * T, -1), L["w_ih"], L["b"]).view(R, T, 2, 4 * self.hidden)
            rec = K.bilstm_recurrence(xg, L["w_hh"], self.hidden, mode=rnn if rnn is not None else K.routing_of(L["w_ih"]).rnn,
                                      status=status)
            x = K.linear(rec.view(R * T, -1), L["lin_w"], L["lin_b"]).view(R, T, -1)
        return x

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        """the reference's call surface (NCHW in): no later read-back to ride on, so the hand-off status is read here (one
        small synchronising copy) and a give-up re-runs the layer stack on the step kernels"""
        from ..backbone.resnet_fpn import as_nhwc
        x = as_nhwc(features)
        guard = K.HandoffGuard(x.device, owner=self)
        out = self.forward_nhwc(x, rnn=getattr(self, "rnn_override", None), status=guard.status)
        guard.retry = lambda: self.forward_nhwc(x, rnn="steps")
        again = guard.resolve(guard.status.item())
        return out if again is None else again

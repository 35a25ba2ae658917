"""CPU: pin the oracle restatement (oracle/glass_cpu.py) against the golden vectors produced
by the reference's own modules (oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from glass_amd.utils.synth import make_state_dict
from oracle import glass_cpu as O


@pytest.fixture(scope="module")
def sd():
    return make_state_dict(1234, parts=("recog",))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_local_extractor(sd, golden_dir):
    g = _load(golden_dir, "local_extractor.npz")
    y = O.local_extractor(sd, torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)


def test_fusion_attention(sd, golden_dir):
    g = _load(golden_dir, "fusion_attention.npz")
    y = O.gc_attention_fusion(sd, torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)


def test_p2p3(sd, golden_dir):
    g = _load(golden_dir, "p2p3_fusion.npz")
    y = O.p2p3_fusion(sd, torch.from_numpy(g["p2"]), torch.from_numpy(g["p3"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-5)


def test_bilstm(sd, golden_dir):
    g = _load(golden_dir, "bilstm_encoder.npz")
    y = O.bilstm_encoder(sd, torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-5)


def test_decoder_and_early_break(sd, golden_dir):
    g = _load(golden_dir, "attention_decoder.npz")
    x = torch.from_numpy(g["x"])
    y = O.attention_decoder(sd, x)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-5)
    k = "roi_heads.recognizer_head.decoder.recognizer.decoder.fc.bias"
    sd2 = dict(sd)
    sd2[k] = sd[k].clone()
    sd2[k][0] += float(g["bias0_break0"])
    y2 = O.attention_decoder(sd2, x)
    np.testing.assert_allclose(y2.numpy(), g["y_break0"], rtol=0, atol=1e-5)
    assert (y2[:, 1:] == 0).all()
    sd3 = dict(sd)
    sd3[k] = sd[k].clone()
    sd3[k][0] += float(g["bias0_partial"])
    y3 = O.attention_decoder(sd3, torch.from_numpy(g["x_partial"]))
    np.testing.assert_allclose(y3.numpy(), g["y_partial"], rtol=0, atol=1e-5)
    assert (y3[:, 9:] == 0).all() and (y3[:, 8].sum(-1) > 0).all()


def test_recognizer_cnn_oracle_matches_reference_module(golden_dir):
    """a9: oracle.recognizer_cnn vs the reference's CNN_V1_1 run with eval-mode BN (make_golden --variants)."""
    import numpy as np
    import os
    import torch
    from oracle import glass_cpu as O
    g = np.load(os.path.join(golden_dir, "recognizer_cnn_variants.npz"))
    pre = "roi_heads.recognizer_head.backbone."
    sd = {pre + k[len("CNN_V1_1:"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("CNN_V1_1:") and not k.endswith(":y")}
    y = O.recognizer_cnn(sd, torch.from_numpy(g["x"]))
    assert float((y - torch.from_numpy(g["CNN_V1_1:y"])).abs().max()) < 2e-5


def test_oracle_fp16_emulation_is_fold_then_round():
    """oracle.glass_cpu.emulate: conv (+ eval BatchNorm) in the emulated arithmetic = BatchNorm folded into the weights in
    float64 (glass_amd/checkpoint.py fold_conv), weights and input rounded to fp16, fp32 accumulation; plain mode is the
    reference's conv2d + batch_norm, untouched; "fp16s" additionally rounds what the product stores as fp16."""
    import torch
    import torch.nn.functional as F
    from oracle import glass_cpu as O
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 8, 9, 7), generator=g)
    sd = {"c.weight": torch.randn((5, 8, 3, 3), generator=g) * 0.2,
          "n.weight": torch.rand((5,), generator=g) + 0.5, "n.bias": torch.randn((5,), generator=g),
          "n.running_mean": torch.randn((5,), generator=g), "n.running_var": torch.rand((5,), generator=g) + 0.5}
    plain = O.conv_bn(x, sd, "c.weight", None, "n", padding=1)
    ref = F.batch_norm(F.conv2d(x, sd["c.weight"], padding=1), sd["n.running_mean"], sd["n.running_var"], sd["n.weight"], sd["n.bias"],
                       training=False, eps=1e-5)
    assert torch.equal(plain, ref)
    scale = (sd["n.weight"].double() / torch.sqrt(sd["n.running_var"].double() + 1e-5))
    wf = (sd["c.weight"].double() * scale.view(-1, 1, 1, 1)).float()
    bf = (sd["n.bias"].double() - sd["n.running_mean"].double() * scale).float()
    want = F.conv2d(x.half().float(), wf.half().float(), bf, padding=1)
    for mode in ("fp16", "fp16s"):
        with O.emulate(mode):
            got = O.conv_bn(x, sd, "c.weight", None, "n", padding=1)
            assert torch.equal(got, want)
            stored = O._st(got)
        assert torch.equal(stored, want.half().float() if mode == "fp16s" else want)
    assert O._EMU["mode"] is None and float((want - ref).abs().max()) > 1e-4       # the emulation really is a different function
    with O.emulate("fp16"):
        y = O.lin(x.reshape(-1, 7), torch.ones((3, 7)) / 3, None)
    assert torch.equal(y, F.linear(x.reshape(-1, 7).half().float(), (torch.ones((3, 7)) / 3).half().float()))

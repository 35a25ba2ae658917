"""The one-launch-per-layer recurrent kernels (csrc/recurrent_persistent.hip) against the one-launch-per-step kernels
(csrc/recognition.hip) they replace: the arithmetic is the same instruction sequence, so the outputs must be BIT-identical -
any difference is a hand-off bug (a stale or torn granule), which a tolerance would hide.  The step kernels themselves are
held to the reference goldens and the oracle in tests/test_gpu_a_stages.py (those tests now run the persistent path, which is
the default routing; reference: glass/modeling/recognition/recognizer_encoder.py:118-144, prediction_aster.py:63-99).

Hand-offs are exercised the way cdna_hip_programming.md Guideline 16 asks: under UNEVEN load (a convolution stream competing
for the CUs, workgroups of a set starting at different times), many repetitions, every output word compared."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _lstm_inputs(R, T, seed):
    g = torch.Generator().manual_seed(seed)
    xg = (torch.randn((R, T, 2, 1024), generator=g) * 1.5).to(_dev())
    whh = (torch.randn((2, 1024, 256), generator=g) * 0.08).to(_dev())
    return xg, whh


@pytest.mark.parametrize("mode", [(2, 1), (2, 2), (1, 1)])
@pytest.mark.parametrize("R,T", [(1, 32), (16, 32), (37, 32), (256, 32), (530, 7), (1040, 32)])
def test_persistent_bilstm_is_bit_identical_to_the_step_kernels(R, T, mode):
    from glass_amd.ops import native as K
    xg, whh = _lstm_inputs(R, T, 7 * R + T)
    ref = K.bilstm_recurrence(xg, whh, 256, mode="steps")
    for rep in range(3):
        got = K.bilstm_recurrence(xg, whh, 256, mode=mode)
        assert torch.equal(got, ref), f"rep {rep}: max diff {float((got - ref).abs().max())}"
    assert K.recurrence_status() == 0


def test_persistent_bilstm_under_uneven_load():
    """a stream of full-chip convolutions next to the recurrent layer: the workgroups of a set start at different times, some
    only when a convolution workgroup retires - every output word must still be the step kernels'"""
    from glass_amd.ops import native as K
    dev = _dev()
    xg, whh = _lstm_inputs(256, 32, 11)
    xg2, _ = _lstm_inputs(48, 32, 12)
    ref, ref2 = K.bilstm_recurrence(xg, whh, 256, mode="steps"), K.bilstm_recurrence(xg2, whh, 256, mode="steps")
    x = torch.randn((8, 128, 128, 256), device=dev)
    w = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp32")
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    bad = 0
    for rep in range(30):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 4):
                K.conv2d_nhwc(x, w, None, padding=1)
        for mode in ((2, 1), (1, 1), (2, 2)):
            a = K.bilstm_recurrence(xg, whh, 256, mode=mode)
            b = K.bilstm_recurrence(xg2, whh, 256, mode=mode)
            bad += int(not torch.equal(a, ref)) + int(not torch.equal(b, ref2))
    torch.cuda.synchronize()
    assert bad == 0
    assert K.recurrence_status() == 0


def test_encoder_module_takes_the_persistent_path_by_default():
    from glass_amd.ops import native as K
    assert K.Routing().rnn == (1, 1) and K.Routing(rnn="persistent").rnn == "persistent" and K.Routing(rnn="steps").rnn == "steps" and K.Routing(rnn="2x2").rnn == (2, 2)


def _decoder(cfg_path_args=()):
    import os
    from glass_amd.config import get_glass_cfg
    from glass_amd.modeling.recognition.recognizer_decoder import ASTER_V2
    from glass_amd.structures.core import ShapeSpec
    from glass_amd.utils.synth import make_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_glass_cfg(os.path.join(root, "configs", "glass_icdar15_mi355x.yaml"))
    sd = make_state_dict(1234)
    dec = ASTER_V2(cfg, ShapeSpec(channels=256))
    dec.import_weights(sd, _dev(), "roi_heads.recognizer_head.decoder.")
    return dec, sd


@pytest.mark.parametrize("R", [1, 16, 37, 256, 530])
def test_persistent_decoder_matches_the_step_kernels(R):
    """glass_attention_decode_persistent vs glass_attention_decode on the same inputs: the same arithmetic in another summation
    order (K-quarters of the GRU, two halves of sEmbed, the embedding half as a table), so probabilities agree to 1e-5 wherever
    the greedy arg-max took the same branch; RoIs whose two best classes are closer than fp32 rounding at some step are
    compared up to that step (as tests/test_gpu_a_stages.py does against the oracle).  Includes the per-image early break."""
    import numpy as np
    from glass_amd.ops import native as K
    dec, _ = _decoder()
    g = torch.Generator().manual_seed(300 + R)
    x = torch.randn((R, 32, 256), generator=g).to(_dev())
    cuts = [R // 5, R // 5 + R // 2]
    ri = torch.zeros((R,), dtype=torch.int32)
    ri[cuts[0]:cuts[1]] = 1
    ri[cuts[1]:] = 2
    ri = ri.to(_dev())
    xproj = K.linear(x.view(R * 32, 256), dec.w["xW"], dec.w["xB"]).view(R, 32, 256)
    ref = K.attention_decode(x, xproj, dec.w, ri, 3, dec.num_classes, dec.max_word_len, 0, mode="steps").cpu().numpy()
    for rep in range(3):
        got = K.attention_decode(x, xproj, dec.w, ri, 3, dec.num_classes, dec.max_word_len, 0, mode=(1, 1)).cpu().numpy()
        assert K.recurrence_status() == 0
        srt = np.sort(ref, axis=-1)
        near_tie = (srt[..., -1] - srt[..., -2]) < 1e-5
        live = ref.sum(-1) > 0
        first_tie = np.where((near_tie & live).any(1), (near_tie & live).argmax(1), 26)
        mask = np.arange(26)[None, :] <= first_tie[:, None]
        assert (first_tie < 26).mean() < 0.02
        err = float(np.abs(got - ref)[mask].max())
        assert err < 1e-5, (rep, err)
        assert ((got.sum(-1) == 0) == (ref.sum(-1) == 0))[first_tie == 26].all(), "early-break zero rows differ"


def test_persistent_decoder_under_uneven_load():
    import numpy as np
    from glass_amd.ops import native as K
    dec, _ = _decoder()
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    x = torch.randn((256, 32, 256), generator=g).to(dev)
    ri = (torch.arange(256) // 32).to(torch.int32).to(dev)
    xproj = K.linear(x.view(256 * 32, 256), dec.w["xW"], dec.w["xB"]).view(256, 32, 256)
    ref = K.attention_decode(x, xproj, dec.w, ri, 8, dec.num_classes, dec.max_word_len, 0, mode=(1, 1))
    xc = torch.randn((8, 128, 128, 256), device=dev)
    wc = K.prepare_conv_weights(torch.randn((256, 3, 3, 256), device=dev) * 0.05, "fp32")
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    bad = 0
    for rep in range(20):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 4):
                K.conv2d_nhwc(xc, wc, None, padding=1)
        got = K.attention_decode(x, xproj, dec.w, ri, 8, dec.num_classes, dec.max_word_len, 0, mode=(1, 1))
        bad += int(not torch.equal(got, ref))          # the same kernel on the same inputs: bit-identical whatever the timing
    torch.cuda.synchronize()
    assert bad == 0 and K.recurrence_status() == 0


@pytest.mark.parametrize("T,R", [(20, 24), (7, 5)])
def test_persistent_decoder_with_fewer_than_32_time_steps(T, R):
    """T < 32 encoder positions (narrower crops): the step-invariant rows beyond T are zero-filled in LDS and masked out of the
    attention soft-max - the persistent kernel against the step kernels."""
    import numpy as np
    from glass_amd.ops import native as K
    dec, _ = _decoder()
    g = torch.Generator().manual_seed(T * 100 + R)
    x = torch.randn((R, T, 256), generator=g).to(_dev())
    ri = torch.zeros((R,), dtype=torch.int32, device=_dev())
    xproj = K.linear(x.view(R * T, 256), dec.w["xW"], dec.w["xB"]).view(R, T, 256)
    ref = K.attention_decode(x, xproj, dec.w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode="steps").cpu().numpy()
    got = K.attention_decode(x, xproj, dec.w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode=(1, 1)).cpu().numpy()
    assert K.recurrence_status() == 0
    srt = np.sort(ref, axis=-1)
    near_tie = (srt[..., -1] - srt[..., -2]) < 1e-5
    live = ref.sum(-1) > 0
    first_tie = np.where((near_tie & live).any(1), (near_tie & live).argmax(1), 26)
    mask = np.arange(26)[None, :] <= first_tie[:, None]
    assert float(np.abs(got - ref)[mask].max()) < 1e-5


def test_decoder_shapes_the_persistent_kernel_does_not_take_run_the_step_kernels():
    """T > 32 or C > 128 are outside the persistent kernel's envelope (LDS rows, class slices): the wrapper must fall back to the
    step kernels, not fail and not produce garbage"""
    from glass_amd._lib import lib
    assert lib().glass_decode_persistent_supported(32, 256, 97, 26) == 1
    assert lib().glass_decode_persistent_supported(40, 256, 97, 26) == 0 and lib().glass_decode_persistent_supported(32, 256, 200, 26) == 0
    from glass_amd.ops import native as K
    dec, _ = _decoder()
    x = torch.randn((6, 40, 256), generator=torch.Generator().manual_seed(5)).to(_dev())
    ri = torch.zeros((6,), dtype=torch.int32, device=_dev())
    xproj = K.linear(x.view(6 * 40, 256), dec.w["xW"], dec.w["xB"]).view(6, 40, 256)
    a = K.attention_decode(x, xproj, dec.w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode=(1, 1))
    b = K.attention_decode(x, xproj, dec.w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode="steps")
    assert torch.equal(a, b)


def test_persistent_decoder_honours_the_temperature():
    """`output = self.fc(output) * self.temperature` (reference prediction_aster.py DecoderUnit.forward): a temperature other than
    1 changes the probabilities (not the arg-max) - same results as the step kernels"""
    import numpy as np
    from glass_amd.ops import native as K
    dec, _ = _decoder()
    w = dict(dec.w)
    w["temperature"] = 0.6
    x = torch.randn((20, 32, 256), generator=torch.Generator().manual_seed(9)).to(_dev())
    ri = torch.zeros((20,), dtype=torch.int32, device=_dev())
    xproj = K.linear(x.view(20 * 32, 256), dec.w["xW"], dec.w["xB"]).view(20, 32, 256)
    a = K.attention_decode(x, xproj, w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode=(1, 1)).cpu().numpy()
    b = K.attention_decode(x, xproj, w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode="steps").cpu().numpy()
    c = K.attention_decode(x, xproj, dec.w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode=(1, 1)).cpu().numpy()
    live = b.sum(-1) > 0
    assert float(np.abs(a - b)[live].max()) < 1e-5 and float(np.abs(a - c).max()) > 1e-4

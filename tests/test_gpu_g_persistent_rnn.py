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

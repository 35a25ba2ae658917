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


# ------------------------------------------------------------------------------------------- fail-safe
# A persistent kernel whose bounded wait gives up returns garbage.  The product path must never hand that out: every
# persistent launch gets a per-call status word, the owner of the step's next read-back resolves it (ops.native.HandoffGuard)
# and re-runs the recognizer's encoder / decoder on the step kernels.  glass_recurrence_test_hook forces the give-up: the
# workgroup that draws start ticket 0 never publishes, and its peers give up after a few thousand sweeps instead of ~2-4 s.

class _forced_giveup:
    def __init__(self, spin_limit=3000, withhold=0):
        self.args = (spin_limit, withhold)

    def __enter__(self):
        from glass_amd.ops import native as K
        K.recurrence_test_hook(*self.args)

    def __exit__(self, *a):
        from glass_amd.ops import native as K
        K.recurrence_test_hook(0, -1)
        torch.cuda.synchronize()
        K.recurrence_status(reset=True)


def test_forced_giveup_raises_the_call_status_and_the_sticky_word():
    """kernel level: the hook makes both persistent kernels give up; bit 0 (BiLSTM) / bit 1 (decoder) land in the caller's
    per-call word AND in the device's sticky word; a launch without the hook leaves both at 0 and is correct again"""
    from glass_amd.ops import native as K
    xg, whh = _lstm_inputs(40, 32, 5)
    ref = K.bilstm_recurrence(xg, whh, 256, mode="steps")
    dec, _ = _decoder()
    x = torch.randn((40, 32, 256), generator=torch.Generator().manual_seed(9)).to(_dev())
    ri = torch.zeros((40,), dtype=torch.int32, device=_dev())
    xproj = K.linear(x.view(40 * 32, 256), dec.w["xW"], dec.w["xB"]).view(40, 32, 256)
    K.recurrence_status(reset=True)
    with _forced_giveup():
        st = K.new_handoff_status(_dev())
        K.bilstm_recurrence(xg, whh, 256, mode=(1, 1), status=st)
        assert int(st.item()) == 1
        assert K.recurrence_status(reset=True) == 1
        st2 = K.new_handoff_status(_dev())
        K.attention_decode(x, xproj, dec.w, ri, 1, dec.num_classes, dec.max_word_len, 0, mode=(1, 1), status=st2)
        assert int(st2.item()) == 2
        assert K.recurrence_status(reset=True) == 2
        # a launch WITHOUT a per-call word still reports into the sticky one
        K.bilstm_recurrence(xg, whh, 256, mode=(2, 1))
        assert K.recurrence_status(reset=True) == 1
    st = K.new_handoff_status(_dev())
    got = K.bilstm_recurrence(xg, whh, 256, mode=(1, 1), status=st)
    assert int(st.item()) == 0 and K.recurrence_status() == 0 and torch.equal(got, ref)


def test_reference_surface_modules_fall_back_to_the_step_kernels_on_a_giveup(golden_dir):
    """`encoder(x)` / `decoder(x)` - the reference's call surface, no later read-back - under a forced give-up still return the
    reference goldens (bilstm_encoder.npz, attention_decoder.npz): the status is read inside the call and the step kernels
    re-run it (reference recognizer_encoder.py:118-144, prediction_aster.py:63-99)."""
    import os
    import numpy as np
    import glass_amd
    from glass_amd.config import get_glass_cfg
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import make_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_glass_cfg(os.path.join(root, "configs", "glass_icdar15_mi355x.yaml"))
    m = glass_amd.build_model(cfg)
    m.load_state_dict(make_state_dict(1234))
    enc, dec = m.roi_heads.recognizer_head.encoder, m.roi_heads.recognizer_head.decoder
    ge = np.load(os.path.join(golden_dir, "bilstm_encoder.npz"))
    gd = np.load(os.path.join(golden_dir, "attention_decoder.npz"))
    n0 = K.handoff_giveups()
    with _forced_giveup():
        y = enc(torch.from_numpy(ge["x"]).to(_dev())).cpu().numpy()
        p = dec(torch.from_numpy(gd["x"]).to(_dev())).cpu().numpy()
    assert K.handoff_giveups() == n0 + 2, "both calls must have gone through the fall-back"
    assert float(np.abs(y - ge["y"]).max()) < 1e-4
    assert float(np.abs(p - gd["y"]).max()) < 1e-4
    # and without the hook the persistent path itself is taken (no further give-up is counted)
    y2 = enc(torch.from_numpy(ge["x"]).to(_dev())).cpu().numpy()
    assert K.handoff_giveups() == n0 + 2 and float(np.abs(y2 - ge["y"]).max()) < 1e-4


def test_model_inference_survives_a_dead_handoff_and_goes_sticky_after_three():
    """product level: `model.inference` (with and without its post-process read-back), the pipelined step generator and
    `GlassRunner` under a forced give-up return EXACTLY what a model routed to the step kernels returns; the fall-back is
    counted, and after HandoffGuard.STICKY_AFTER give-ups the head stays on the step kernels (no further give-ups)."""
    import os
    import glass_amd
    from glass_amd.config import get_glass_cfg
    from glass_amd.ops import native as K
    from glass_amd.utils.pipeline import run_pipelined
    from glass_amd.utils.synth import make_boxes, make_image, make_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_glass_cfg(os.path.join(root, "configs", "glass_icdar15_mi355x.yaml"), ["MODEL.DEVICE", "cuda:0"])
    sd = make_state_dict(1234)
    H, W, B, R = 160, 224, 2, 20                     # 40 RoIs: three 16-RoI groups, the last one ragged
    imgs = [make_image(70 + i, H, W).permute(2, 0, 1).float().contiguous().cuda() for i in range(B)]
    boxes = [(make_boxes(70 + i, R, H, W) * torch.tensor([1, 1, 0.3, 0.5, 1.0])).cuda() for i in range(B)]
    inputs = [{"image": im} for im in imgs]

    os.environ["GLASS_RNN"] = "steps"
    try:
        m_steps = glass_amd.build_model(cfg)
    finally:
        del os.environ["GLASS_RNN"]
    m_steps.load_state_dict(sd)
    assert m_steps.routing.rnn == "steps"
    ref_raw = m_steps.inference(inputs, do_postprocess=False, override_boxes=boxes).batch.text.clone()
    ref_out = m_steps.inference(inputs, override_boxes=boxes)
    ref_post = ref_out.batch.text.clone()

    m = glass_amd.build_model(cfg)
    m.load_state_dict(sd)
    assert m.routing.rnn == (1, 1)
    head = m.roi_heads.recognizer_head
    # sanity: untouched, the persistent path differs from the step kernels only by summation order and raises nothing
    n0 = K.handoff_giveups()
    ok = m.inference(inputs, do_postprocess=False, override_boxes=boxes).batch.text
    assert K.handoff_giveups() == n0 and float((ok - ref_raw).abs().max()) < 1e-4
    with _forced_giveup():
        got_raw = m.inference(inputs, do_postprocess=False, override_boxes=boxes).batch.text.clone()       # own status read-back
        assert K.handoff_giveups() == n0 + 1 and head.rnn_override is None
        assert torch.equal(got_raw, ref_raw), "the fall-back must return the step kernels' result bit for bit"
        got = m.inference(inputs, override_boxes=boxes)                                                      # rides on the count read-back
        assert K.handoff_giveups() == n0 + 2 and head.rnn_override is None
        assert torch.equal(got.batch.text, ref_post)
        assert [len(a["instances"]) for a in got] == [len(a["instances"]) for a in ref_out]
        piped = run_pipelined([(lambda: m.inference_g(inputs, override_boxes=boxes)) for _ in range(2)], depth=2, device=_dev())
        torch.cuda.synchronize()
        assert all(torch.equal(p.batch.text, ref_post) for p in piped)
        # third + fourth give-up seen: the head is now pinned to the step kernels ...
        assert head.rnn_override == "steps" and head.rnn_giveups >= K.HandoffGuard.STICKY_AFTER
        n1 = K.handoff_giveups()
        again = m.inference(inputs, override_boxes=boxes)
        # ... so the hook (still armed) finds no persistent launch to break
        assert K.handoff_giveups() == n1 and torch.equal(again.batch.text, ref_post)
    assert K.recurrence_status() == 0

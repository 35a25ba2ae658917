"""CPU: the detectron2 pins.  `scripts/pin_d2.py`, run once where detectron2 v0.6 is importable, stores detectron2's OWN outputs
for the ops the oracle restates from memory (tests/golden/d2_<op>.npz: seeded inputs + detectron2's results).  From then on this
module holds oracle/d2ops.py (and through it every HIP kernel compared with the oracle) to those files - here, on the GPU box,
anywhere.  Until such files exist the d2-owned half stays "parity unpinned"  and the consuming test skips,
saying so; the plumbing (every op has inputs, the oracle runs on them, the dry run lists them) is tested regardless."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

OPS = ["pairwise_iou_rotated", "nms_rotated", "roi_align_rotated", "roi_pooler", "rotated_anchor_generator",
       "box2box_transform_rotated", "rotated_boxes_clip_scale", "find_top_rrpn_proposals"]


def test_pin_script_dry_run_lists_every_op_and_refuses_politely_without_detectron2():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_d2.py"), "--dry-run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for op in OPS:
        assert op in r.stdout and f"tests/golden/d2_{op}.npz" in r.stdout
    try:
        import detectron2  # noqa: F401
    except Exception:      # noqa: BLE001 - the normal case in both images
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_d2.py"), "--out", "/nonexistent"], capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 2 and "not importable" in r.stderr and "parity unpinned" in r.stderr


def test_every_pin_has_inputs_the_oracle_runs_on():
    """the oracle side of each pin executes on the pin's own seeded inputs and returns finite, well-formed results - so the day
    detectron2 is available the only open question is whether the numbers agree"""
    import pin_d2
    pins = pin_d2._pins()
    assert list(pins) == OPS
    for name, p in pins.items():
        ins = p["inputs"]()
        out = p["run_oracle"](ins)
        assert out, name
        for k, v in out.items():
            assert isinstance(v, torch.Tensor) and v.numel() >= 0, (name, k)
            if v.dtype.is_floating_point and name != "find_top_rrpn_proposals":
                assert torch.isfinite(v).all(), (name, k)
        # the comparison routine accepts the oracle against itself and rejects a perturbed copy
        ref = {k: v.numpy() for k, v in out.items()}
        assert pin_d2.compare(name, ref, ref, p["tol"]) == []
        k0 = next(k for k, v in ref.items() if v.size)
        pert = dict(ref)
        pert[k0] = ref[k0] + (1 if ref[k0].dtype.kind in "iu" else 1e-2)
        assert pin_d2.compare(name, pert, ref, p["tol"]), name


def test_oracle_reproduces_detectron2_on_the_pinned_ops():
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "d2_*.npz")))
    if not files:
        pytest.skip("no tests/golden/d2_*.npz: detectron2 was never importable where scripts/pin_d2.py ran - the d2-owned half of the "
                    "oracle is pinned by analytic known answers only (tests/test_oracle_d2ops.py)")
    import pin_d2
    pins = pin_d2._pins()
    bad = []
    for f in files:
        name = os.path.basename(f)[3:-4]
        g = np.load(f, allow_pickle=False)
        ins = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
        ref = {k[4:]: g[k] for k in g.files if k.startswith("out_")}
        got = {k: v.numpy() for k, v in pins[name]["run_oracle"](ins).items()}
        bad += pin_d2.compare(name, got, ref, pins[name]["tol"])
    assert not bad, "\n".join(bad)

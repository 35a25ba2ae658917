"""Timing ablation (wrong results, timing only): how much of the pipelined step do the recurrent launch chains cost?
    GLASS_ABL_TAIL=lstm|decoder|gc|all|skinny python scripts/exp_tail_ablation.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras
runs bench.py with the named ops replaced by an allocation of their output (no launches)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd")); sys.path.insert(0, ROOT)
import torch
from glass_amd.ops import native as K
what = os.environ.get("GLASS_ABL_TAIL", "")
if what in ("lstm", "all"):
    def _no_lstm(xg, w_hh_packed, hidden):
        return xg[:, :, 0, :2 * hidden].contiguous()          # (input-dependent, one copy kernel: bench.py checks that steps differ)
    K.bilstm_recurrence = _no_lstm
if what in ("decoder", "all"):
    def _no_dec(x, xproj, weights, roi_image, num_images, num_classes, max_len, eos):
        return torch.softmax(x[:, :max_len, :num_classes], -1)
    K.attention_decode = _no_dec
if what in ("gc", "all"):
    K.gc_attention_inplace = lambda x, *a, **k: x
if what in ("skinny",):                    # the few-workgroup, long-K layers: box predictor (Cout 11, K 2048) -> a slice of its input
    _lin = K.linear
    def _lin_abl(x, w, bias=None, relu=0, out=None, out_dtype=None, precision=None):
        n = (w.raw if isinstance(w, K.ConvWeight) else w).shape[0]
        if n <= 16 and out is None:
            return x[:, :n].contiguous()
        return _lin(x, w, bias, relu, out, out_dtype, precision)
    K.linear = _lin_abl
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")

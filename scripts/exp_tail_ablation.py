"""Timing ablation (wrong results, timing only): how much of the pipelined step do the recurrent launch chains cost?
    GLASS_ABL_TAIL=lstm|decoder|gc|all python scripts/exp_tail_ablation.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras
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
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")

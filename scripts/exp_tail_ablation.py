"""Timing ablation (wrong results, timing only): how much of the pipelined step do the recurrent launch chains cost?
    GLASS_ABL_TAIL=lstm|decoder|gc|all|skinny|nms|rpn|det|pp python scripts/exp_tail_ablation.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras
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
if what in ("nms", "det"):                 # rotated NMS (RPN proposals + detections): the first post_topk candidates, no suppression
    def _no_nms(boxes, scores, cat, valid_count, image_hw, score_thresh, nms_thresh, post_topk, flags):
        N, S = scores.shape
        k = min(S, post_topk)
        ob = torch.zeros((N, post_topk, 5), dtype=torch.float32, device=boxes.device); ob[:, :k] = boxes[:, :k]
        os_ = torch.zeros((N, post_topk), dtype=torch.float32, device=boxes.device); os_[:, :k] = scores[:, :k]
        oi = torch.arange(post_topk, dtype=torch.int32, device=boxes.device).repeat(N, 1)
        oc = torch.full((N,), min(k, 100), dtype=torch.int32, device=boxes.device)
        return ob, os_, oi, oc
    K.rotated_nms_select = _no_nms
if what in ("rpn", "det"):                 # RPN top-k select + decode: outputs left as allocated
    K.rpn_topk_decode = lambda *a, **k: None
if what in ("pp",):                        # word post-processor: no survivors
    _pp = K.postprocess_words
    def _no_pp(boxes, scores, counts, text, scale_xy, thr, stop):
        N, KK, _ = boxes.shape
        T = int(text.shape[2]) if text is not None else 1
        dev = boxes.device
        z = lambda *sh, dt=torch.float32: torch.zeros(sh, dtype=dt, device=dev)
        return {"boxes": z(N, KK, 5), "scores": z(N, KK), "polygons": z(N, KK, 4, 2), "src": z(N, KK, dt=torch.int32), "char": z(N, KK, T, dt=torch.int32),
                "text_score": z(N, KK), "text_len": z(N, KK, dt=torch.int32), "count": z(N, dt=torch.int32)}
    K.postprocess_words = _no_pp
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")

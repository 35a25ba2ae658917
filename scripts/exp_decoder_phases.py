"""Phase stamps of the persistent decoder kernel (variant library built with -DGLASS_PL_STAMPS):
   scripts/build_variant_lib.sh plst -DGLASS_PL_STAMPS && GLASS_HIP_LIB=$PWD/glass-text-spotting_amd/libglass_hip_plst.so python scripts/exp_decoder_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd._lib import lib
from glass_amd.ops import native as K
from glass_amd.config import get_glass_cfg
from glass_amd.modeling.recognition.recognizer_decoder import ASTER_V2
from glass_amd.structures.core import ShapeSpec
from glass_amd.utils.synth import make_state_dict
dev = torch.device("cuda:0")
cfg = get_glass_cfg(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "glass_icdar15_mi355x.yaml"))
dec = ASTER_V2(cfg, ShapeSpec(channels=256))
dec.import_weights(make_state_dict(1234), dev, "roi_heads.recognizer_head.decoder.")
L = lib()
names = ["A sweep h_i + barrier", "B W_hh h, sEmbed(h) MFMAs (wavefronts 4-7) | fc class slice MFMAs (0-3) + barrier",
         "publish sEmbed + fc slices; C wait for this RoI's sEmbed row (0-3) | its logits -> soft-max / arg-max / out row (4) + barrier",
         "C energies (tanh) + barrier", "C soft-max + context + barrier", "publish ctx; E wait for the group's contexts",
         "E symbols + embedding rows + barrier", "F context-half MFMAs (wavefronts 0-3) + barrier", "F cell update, publish h"]
for R in (32, 256):
    x = torch.randn((R, 32, 256), device=dev)
    ri = (torch.arange(R) // 32).to(torch.int32).to(dev)
    xp = K.linear(x.view(R * 32, 256), dec.w["xW"], dec.w["xB"]).view(R, 32, 256)
    out = torch.empty((R, 26, 97), device=dev)
    pred = torch.empty((R, 26), dtype=torch.int32, device=dev)
    w = K.DecoderWeights()
    for n in ("sW", "sB", "wW", "wB", "emb", "w_ih", "w_hh", "b_ih", "b_hh", "fcW", "fcB"):
        setattr(w, n, dec.w[n].data_ptr())
    w.temperature = 1.0
    nb = int(L.glass_decode_persistent_workspace_bytes(R))
    ws = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    for _ in range(3):
        rc = L.glass_attention_decode_persistent(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(xp.data_ptr()), ctypes.byref(w),
                                                 ctypes.c_void_p(dec.w["sW_rm"].data_ptr()), ctypes.c_void_p(dec.w["emb_gi"].data_ptr()),
                                                 ctypes.c_void_p(ri.data_ptr()), R, int(ri.max()) + 1, 32, 256, 97, 26, 0, ctypes.c_void_p(out.data_ptr()),
                                                 ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(None), ctypes.c_void_p(ws.data_ptr()), ctypes.c_int64(nb), ctypes.c_void_p(K.stream_handle()))
        assert rc == 0
        torch.cuda.synchronize()
    st = ws[64:64 + 192].view(torch.int64).cpu().tolist()
    print(f"R={R}: shader cycles per decoding step (26 steps), ticket 0: wavefront 0 | wavefront 4")
    for k, nm in enumerate(names):
        print(f"   {st[k] / 26:8.0f} | {st[12 + k] / 26:8.0f}   {nm}")
    print(f"   total {sum(st[:9]) / 26:8.0f} | {sum(st[12:21]) / 26:8.0f} cycles per step")

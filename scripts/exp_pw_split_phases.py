"""Phase stamps of the bf16-split 1x1 kernel (variant library built with -DGLASS_PWS_STAMPS):
   scripts/build_variant_lib.sh pwst -DGLASS_PWS_STAMPS && GLASS_HIP_LIB=$PWD/glass-text-spotting_amd/libglass_hip_pwst.so python scripts/exp_pw_split_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd._lib import lib
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
L = lib()
dbg = torch.zeros((16,), dtype=torch.int64, device=dev)
L.glass_pws_debug(ctypes.c_void_p(dbg.data_ptr()))
names = ["prologue (first loads -> LDS, barrier)", "first B fragments read (LDS latency)", "px-blocks 0..PB-XL-1: MFMA + load issue", "px-blocks PB-XL..: MFMA + split + LDS writes",
         "barrier", "epilogue"]
for (N, H, W, Cin, Cout, st, has_res) in [(8, 256, 256, 256, 256, 1, 1), (8, 64, 64, 1024, 2048, 2, 0), (8, 128, 128, 512, 256, 1, 1)]:
    x = torch.randn((N, H, W, Cin), device=dev)
    w_raw = torch.randn((Cout, 1, 1, Cin), device=dev) * (1.0 / Cin ** 0.5)
    w = K.prepare_conv_weights(w_raw, "fp32")
    w.packs["pws"] = K.winograd_pack(w_raw, "pws")
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    res = torch.randn((N, Ho, Wo, Cout), device=dev) if has_res else None
    y = torch.empty((N, Ho, Wo, Cout), device=dev)
    for force in ("pws9", "pws6"):
        for _ in range(3):
            K.conv2d_nhwc(x, w, None, stride=st, relu=1, residual=res, res_mode=1 if has_res else 0, out=y, winograd=force)
        torch.cuda.synchronize()
        nk = Cin // 32
        for blk in (0, 1):
            v = dbg[8 * blk:8 * blk + 8].cpu().tolist()
            print(f"[{N},{H},{W},{Cin}]->{Cout} s{st} {force} workgroup {blk}: total {sum(v[:6]) * 10} ns; per k-tile (ns): " +
                  "  ".join(f"{names[k].split(':')[0].split('(')[0].strip()} {v[k] * 10 / (nk if 1 <= k <= 4 else 1):.0f}" for k in range(6)))

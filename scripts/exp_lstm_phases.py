"""Phase stamps of the persistent BiLSTM kernel (variant library built with -DGLASS_PL_STAMPS):
   scripts/build_variant_lib.sh plst -DGLASS_PL_STAMPS && GLASS_HIP_LIB=$PWD/glass-text-spotting_amd/libglass_hip_plst.so python scripts/exp_lstm_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd._lib import lib
from glass_amd.ops.native import stream_handle
dev = torch.device("cuda:0")
L = lib()
for R in (32, 256):
    for nd, ng in ((1, 1), (2, 1)):
        T = 32
        xg = torch.randn((R, T, 2, 1024), device=dev)
        whh = torch.randn((2, 1024, 256), device=dev) * 0.08
        out = torch.empty((R, T, 512), device=dev)
        nb = int(L.glass_bilstm_persistent_workspace_bytes(R, 256))
        ws = torch.zeros((nb,), dtype=torch.uint8, device=dev)
        for _ in range(3):
            rc = L.glass_bilstm_recurrence_persistent(ctypes.c_void_p(xg.data_ptr()), ctypes.c_void_p(whh.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                      R, T, 256, nd, ng, ctypes.c_void_p(None), ctypes.c_void_p(ws.data_ptr()), ctypes.c_int64(nb), ctypes.c_void_p(stream_handle()))
            assert rc == 0
            torch.cuda.synchronize()
        st = ws[64:64 + 48].view(torch.int64).cpu().tolist()
        n = (T - 1) * nd
        names = ["sweep (wait + tag check + LDS write)", "barrier 1", "prefetch issue + B reads + 64 MFMA + gates write", "barrier 2", "gate functions + publish + out store"]
        print(f"R={R} chains/workgroup={nd * ng}: s_memtime ticks (100 MHz) per chain-step, wavefront 0 of ticket 0:")
        for k, nm in enumerate(names):
            print(f"   {st[k] / n:8.1f} ticks = {st[k] / n * 10:7.0f} ns   {nm}")
        print(f"   total {sum(st[:5]) / n * 10:7.0f} ns per chain-step")

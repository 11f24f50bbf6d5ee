"""Micro-benchmark of the two conv3x3 weight-gradient kernels (fp32 implicit GEMM vs split-bf16 MFMA) on the tower's shapes."""
import sys, torch
sys.path.insert(0, ".")
from emotiongestures_amd import _lib as L
from emotiongestures_amd.engine import _ptr, _stream
lib = L.load()
DEV = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for (H, W, C) in ((128, 124, 32), (64, 62, 64), (32, 31, 128)):
    x = torch.randn(B, H, W, C, device=DEV); dy = torch.randn(B, H, W, C, device=DEV)
    out = torch.empty(C, 9 * C, device=DEV)
    ws1 = torch.empty(max(int(lib.eg_gemm_tn_workspace_floats(C, 9 * C, B * H * W)), 1), device=DEV)
    ws2 = torch.empty(int(lib.eg_conv3x3_wgrad_mfma_workspace_floats(B, H, W, C, C)), device=DEV)
    fl = 2.0 * 9 * C * C * H * W * B
    for name, fn in (("f32 ", lambda: lib.eg_conv3x3_wgrad(_ptr(x), _ptr(dy), _ptr(out), B, H, W, C, C, 1, _ptr(ws1), ws1.numel(), _stream(DEV))),
                     ("mfma", lambda: lib.eg_conv3x3_wgrad_mfma(_ptr(x), _ptr(dy), _ptr(out), B, H, W, C, C, _ptr(ws2), ws2.numel(), _stream(DEV)))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): assert fn() == 0
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"B={B} {H}x{W}x{C} {name}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s  (x+dy {2 * x.numel() * 4 / us * 1e-6:.2f} TB/s)")

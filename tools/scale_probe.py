import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from emotiongestures_amd import _lib as L, ops
from emotiongestures_amd.engine import _ptr, _stream
dev = torch.device("cuda:0"); lib = L.load()
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for prec in ("bf16x3", "bf16"):
    pc = L.precision_code(prec)
    for (cin, cout, H, W) in [(128, 128, 32, 31), (64, 64, 64, 62), (32, 32, 128, 124)]:
        for B in (8, 16, 32, 64, 128, 256):
            x = torch.randn(B, H, W, cin, device=dev); w = torch.randn(cout, cin, 3, 3) * 0.05
            wp, opad = ops.pack_conv3x3_weight(w, dev); y = torch.empty(B, H, W, cout, device=dev); st = _stream(dev)
            us = timeit(lambda: lib.eg_conv3x3(_ptr(x), _ptr(wp), None, None, None, _ptr(y), None, B, H, W, cin, cout, 1, 1, 0, pc, st))
            print(f"{prec} {cin}->{cout} B={B:4d}: {us:8.1f} us  {2.0*9*cin*cout*H*W*B/us/1e6:7.1f} TF")

#!/usr/bin/env python3
"""Micro-benchmarks of the contraction kernels through the C ABI (GPU box).  usage: bench_ops.py [gemm|conv] [prec]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emotiongestures_amd import _lib as L
from emotiongestures_amd import ops
from emotiongestures_amd.engine import _ptr, _stream

dev = torch.device("cuda:0")
lib = L.load()


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def gemm(prec):
    pc = L.precision_code(prec)
    shapes = [(2048, 512, 512), (2048, 512, 2048), (2048, 2048, 512), (2048, 1536, 512), (2176, 512, 512), (2176, 1536, 512), (2176, 2048, 512), (2176, 512, 2048), (2176, 512, 992), (3840, 300, 300),
              (8704, 512, 512), (2176, 128, 512), (69632, 512, 512), (69632, 2048, 512), (69632, 512, 2048),
              (544, 512, 512), (544, 2048, 512), (544, 512, 2048), (4352, 512, 512), (4352, 1536, 512), (4352, 2048, 512), (4352, 512, 2048)]   # training rows: 16 / 128 clips x 34 frames
    pad = int(os.environ.get("LDA_PAD", "0"))
    if os.environ.get("TRAIN_ONLY") == "1":          # the fp32-input product at the training step's row counts only (PMC passes)
        shapes = [sh for sh in shapes if sh[0] in (544, 4352)]
    for (M, N, K) in shapes:
        x = torch.randn(M, K + pad, device=dev)
        w = torch.randn(N, K) * 0.05
        wp, npad, kpad = ops.pack_linear_weight(w, dev)
        y = torch.empty(M, N, device=dev)
        st = _stream(dev)
        f = lambda: lib.eg_linear(_ptr(x), K + pad, _ptr(wp), kpad, None, None, None, 0, _ptr(y), N, M, N, K, 0, 0, 0, pc, st)
        us = timeit(f)
        print(f"gemm {prec:7s} M={M:5d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s (algorithmic)")
        if pc != 0 and pad == 0 and os.environ.get("TRAIN_ONLY") != "1":
            kp, mt = (K + 63) // 64 * 64, (M + 63) // 64
            img = torch.empty(2 * mt * 64 * kp, dtype=torch.int16, device=dev)
            xc = x.contiguous()
            us_s = timeit(lambda: lib.eg_split_tiles(_ptr(xc), K, M, K, _ptr(img), st))
            y2 = torch.empty(M, N, device=dev)
            f(); torch.cuda.synchronize()
            line = f"     presplit: split {us_s:6.1f} us |"
            for tile in (os.environ.get("TILES", "64,64r8,128x64,128x64r3,128x64r6,128,128r4").split(",")):
                os.environ["EG_GEMM_TILE"] = tile
                us_p = timeit(lambda: lib.eg_linear_presplit(_ptr(img), K, _ptr(wp), kpad, None, None, None, 0, _ptr(y2), N, M, N, K, 0, pc, st))
                torch.cuda.synchronize()
                ok = "" if torch.equal(y2, y) else " MISMATCH"
                line += f" {tile}: {us_p:6.1f} us {2.0 * M * N * K / us_p / 1e6:6.1f} TF{ok} |"
            os.environ.pop("EG_GEMM_TILE", None)
            print(line)


def conv(prec):
    pc = L.precision_code(prec)
    B = 64
    torch.manual_seed(0)
    for (cin, cout, s, H, W) in [(32, 32, 1, 128, 124), (64, 64, 1, 64, 62), (128, 128, 1, 32, 31), (32, 64, 2, 128, 124), (64, 128, 2, 64, 62)]:
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(cout, cin, 3, 3) * 0.05
        wp, opad = ops.pack_conv3x3_weight(w, dev)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        y = torch.empty(B, Ho, Wo, cout, device=dev)
        st = _stream(dev)
        f = lambda: lib.eg_conv3x3(_ptr(x), _ptr(wp), None, None, None, _ptr(y), None, B, H, W, cin, cout, s, 1, 0, pc, st)
        us = timeit(f)
        fl = 2.0 * 9 * cin * cout * Ho * Wo * B
        by = 4.0 * B * (H * W * cin + Ho * Wo * cout)
        chk = (float(y.double().sum()), float(y.double().abs().sum()), float(y[B // 2, Ho // 2, Wo // 3, cout // 2]))   # equal across tilings (same summation order)
        print(f"conv {prec:7s} {cin:3d}->{cout:3d} s{s} {H}x{W}: {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s  {by / us / 1e3:7.1f} GB/s (in+out)  checksum {chk[0]:.6e} {chk[1]:.6e} {chk[2]:.6e}")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    for prec in (sys.argv[2:] or ["bf16x3"]):
        (gemm if what == "gemm" else conv)(prec)

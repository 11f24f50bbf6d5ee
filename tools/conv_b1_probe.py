"""Probe: isolated launch time of the tower's stride-1 convolutions at small batches, weights hot (the same layer again and again: its 1.2 MB image stays in L2)
versus cold (cycling through 40 different weight images = 47 MB, more than the 8 x 4 MB of L2).  Usage: python tools/conv_b1_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emotiongestures_amd import ops

dev = torch.device("cuda:0")
for (c, h, w) in ((128, 32, 31), (64, 64, 62), (32, 128, 124)):
    for B in (1, 4, 8, 16, 32, 64):
        x = torch.randn(B, h, w, c, device=dev)
        ws = [torch.randn(c, c, 3, 3, device=dev) * 0.05 for _ in range(40)]
        packs = [ops.conv3x3_pack(wt, None, None, None, dev) for wt in ws]
        for mode in ("hot", "cold"):
            def run(i):
                k = 0 if mode == "hot" else i % 40
                return ops.conv3x3(x, ws[k], precision="bf16x3", packed=packs[k])
            for i in range(40):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(40):
                    y = run(i)
            g.replay(); torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            print(f"C={c} {h}x{w} B={B:3d} {mode}: {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us per launch", flush=True)

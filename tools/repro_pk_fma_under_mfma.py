#!/usr/bin/env python3
"""Concurrency check: the attention kernel's output must not depend on what else runs on the GPU.

Round-1 finding (MI355X, ROCm 7.2): with the P.V loop written on float4s, hipcc emitted
    ds_read_b128 x10 ; s_waitcnt lgkmcnt(7) ; v_pk_fma_f32 ... op_sel_hi:[1,0,1] ; s_waitcnt lgkmcnt(6) ; v_pk_fma_f32 ... op_sel:[0,1,0] ; ...
(packed fp32 FMAs broadcasting one probability out of an LDS-loaded register pair, released one counted lgkmcnt at a time).
Alone on the GPU the kernel was bit-reproducible and correct; while a 128->128 MFMA convolution was resident on the same CUs,
about half of its launches returned one wrong component (columns k, k+4, ..., k+60 of one (row, head): 16 lanes = one
quarter-wave pass of one packed instruction, off by one product term).  The victim needed nothing but its own static inputs;
f32 GEMMs, LayerNorm and the convolutions themselves were unaffected.  Writing the four FMAs as scalar v_fmac_f32 (misc.hip)
removed it: 0 / 800 launches.  This script keeps the check: it prints the number of attention launches whose output differs
from the quiet reference while convolutions run on a second stream (expected: 0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emotiongestures_amd import _lib as L
from emotiongestures_amd import ops
from emotiongestures_amd.engine import _ptr, _stream

dev = torch.device("cuda:0")
lib = L.load()
pcn = L.precision_code("bf16x3")
x = torch.randn(16, 32, 31, 128, device=dev)
wp, _ = ops.pack_conv3x3_weight(torch.randn(128, 128, 3, 3) * 0.05, dev)
y = torch.empty(16, 32, 31, 128, device=dev)


def noise():
    lib.eg_conv3x3(_ptr(x), _ptr(wp), None, None, None, _ptr(y), None, 16, 32, 31, 128, 128, 1, 1, 0, pcn, _stream(dev))


B_, Lq = 4, 34
torch.manual_seed(1)
qkv = torch.randn(B_ * Lq, 1536, device=dev)


def att():
    o = torch.empty(B_ * Lq, 512, device=dev)
    lib.eg_attention(_ptr(qkv), 1536, _ptr(qkv[:, 512:]), 1536, _ptr(qkv[:, 1024:]), 1536, _ptr(o), 512, None, B_, 8, Lq, Lq, 64, L.precision_code("bf16x3"), _stream(dev))
    return o


ref = att().clone()
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
bad = total = 0
for rep in range(20):
    with torch.cuda.stream(sB):
        for _ in range(120):
            noise()
    with torch.cuda.stream(sA):
        outs = [att() for _ in range(40)]
    torch.cuda.synchronize()
    bad += sum(int(not torch.equal(o, ref)) for o in outs)
    total += len(outs)
print(f"attention launches differing from the quiet reference while 128->128 convolutions run: {bad}/{total}")
sys.exit(1 if bad else 0)

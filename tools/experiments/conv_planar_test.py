"""Unit test of the rejected planar first stage (tools/experiments/conv_planar.hip).  NOT collected by pytest (not under tests/); needs the
library rebuilt with conv_planar.hip and its ABI entries restored -- see README.md in this directory."""
import pytest
import torch

from tests.conftest import build_mirror, rel_l2  # noqa: F401

# ---- first tower stage on producer-split activations (csrc/conv_planar.hip) ------------------------------------------------------------
def _to_planes(x_nhwc):
    from emotiongestures_amd import _lib as L
    from emotiongestures_amd.engine import _ptr, _stream
    B, H, W, C = x_nhwc.shape
    p = torch.empty(B * H * W * 32, dtype=torch.float32, device=x_nhwc.device)          # P32: the same bytes as fp32 NHWC at 32 channels
    L.check(L.load().eg_nhwc_to_planar32(_ptr(x_nhwc), _ptr(p), B, H, W, _stream(x_nhwc.device)), "eg_nhwc_to_planar32")
    return p


def _from_planes(p, B, H, W):
    from emotiongestures_amd import _lib as L
    from emotiongestures_amd.engine import _ptr, _stream
    y = torch.empty(B, H, W, 32, dtype=torch.float32, device=p.device)
    L.check(L.load().eg_planar32_to_nhwc(_ptr(p), _ptr(y), B, H, W, _stream(p.device)), "eg_planar32_to_nhwc")
    return y


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("B,H,W", [(2, 24, 40), (3, 19, 70), (2, 128, 124), (1, 5, 9)])
def test_planar_c32_stage_matches_the_nhwc_kernels(B, H, W, prec):
    """The P32 path (stem -> conv1 + pooling partials -> gate from conv1's moments -> conv2 with the fused SE tail, reading and writing bf16 (hi, lo)
    planes, halos staged by LDS-DMA) against the fp32-NHWC kernels on the same data: same products in the same order, so the results agree
    to the planes' own resolution (hi + lo carries 16 mantissa bits: 2^-17 relative per stored activation).  Ragged maps (partial tiles in
    both directions, a map smaller than one tile) included."""
    from emotiongestures_amd import _lib as L
    from emotiongestures_amd import ops
    from emotiongestures_amd.engine import _ptr, _stream
    lib = L.load()
    d = dev()
    pc = L.precision_code(prec)
    st = _stream(d)
    spec = T("spec", (B, H, W), -80, 0).to(d)
    w9 = T("stem.w", (9, 32), -0.05, 0.05).to(d)
    sb, ss, sh = T("stem.b", (32,), -0.2, 0.2).to(d), T("stem.s", (32,), 0.5, 1.5).to(d), T("stem.t", (32,), -0.3, 0.3).to(d)
    # stem
    x_ref = torch.empty(B, H, W, 32, device=d)
    L.check(lib.eg_stem_conv(_ptr(spec), _ptr(w9), _ptr(sb), _ptr(ss), _ptr(sh), _ptr(x_ref), B, H, W, 32, st), "eg_stem_conv")
    xp = torch.empty(B * H * W * 32, device=d)
    L.check(lib.eg_stem_conv_planar(_ptr(spec), _ptr(w9), _ptr(sb), _ptr(ss), _ptr(sh), _ptr(xp), B, H, W, st), "eg_stem_conv_planar")
    assert rel_l2(_from_planes(xp, B, H, W).cpu().numpy(), x_ref.cpu().numpy()) < 1e-5
    # from here on both sides start from the same activations: the planes' values
    x0 = _from_planes(xp, B, H, W)
    w1 = T("c1.w", (32, 32, 3, 3), -0.1, 0.1)
    w2 = T("c2.w", (32, 32, 3, 3), -0.1, 0.1)
    s1, t1 = T("c1.s", (32,), 0.5, 1.5).to(d), T("c1.t", (32,), -0.3, 0.3).to(d)
    s2, t2 = T("c2.s", (32,), 0.5, 1.5).to(d), T("c2.t", (32,), -0.3, 0.3).to(d)
    wp1, _ = ops.pack_conv3x3_weight(w1, d)
    wp2, _ = ops.pack_conv3x3_weight(w2, d)
    # conv1: ReLU -> BN affine, pooling partials
    tiles_ref = int(lib.eg_conv3x3_gap_tiles(H, W, 32, 32, 1))
    y1_ref, gap_ref = torch.empty(B, H, W, 32, device=d), torch.empty(B, tiles_ref, 32, device=d)
    L.check(lib.eg_conv3x3(_ptr(x0), _ptr(wp1), None, _ptr(s1), _ptr(t1), _ptr(y1_ref), _ptr(gap_ref), B, H, W, 32, 32, 1, 1, 0, pc, st), "eg_conv3x3")
    tiles_p = int(lib.eg_conv3x3_c32_planar_gap_tiles(H, W))
    y1p, gap_p = torch.empty(B * H * W * 32, device=d), torch.empty(B, tiles_p, 32, device=d)
    L.check(lib.eg_conv3x3_c32_planar(_ptr(xp), _ptr(wp1), None, _ptr(s1), _ptr(t1), None, None, _ptr(y1p), None, _ptr(gap_p), B, H, W, 1, pc, st),
            "eg_conv3x3_c32_planar")
    y1 = _from_planes(y1p, B, H, W)
    # bf16x3: both sides multiply the same (hi, lo) pairs.  bf16 (one term): the fp32 kernel rounds hi + lo to bf16 again while the planes
    # hand over hi itself, which differs on near-ties -- a 2^-9 step on a few operands, far below the mode's own error vs fp32 (~2e-3)
    tol = 1e-5 if prec == "bf16x3" else 5e-4
    assert rel_l2(y1.cpu().numpy(), y1_ref.cpu().numpy()) < tol
    assert rel_l2(gap_p.sum(1).cpu().numpy(), gap_ref.sum(1).cpu().numpy()) < tol
    # gate from conv1's moments: planes vs fp32 (fed the planes' values, so only the summation structure differs)
    f = lambda k, n, lo, hi: T(k, (n,), lo, hi).to(d)
    fw1, fb1, fw2, fb2 = T("se.w1", (4, 32), -0.3, 0.3).to(d), f("se.b1", 4, -0.1, 0.1), T("se.w2", (32, 4), -0.3, 0.3).to(d), f("se.b2", 32, -0.1, 0.1)
    gate_ref, gate_p = torch.empty(B, 32, device=d), torch.empty(B, 32, device=d)
    if H > 1 and W > 1:
        L.check(lib.eg_se_gate_pre(_ptr(y1), _ptr(gap_p), tiles_p, _ptr(wp2), _ptr(s2), _ptr(t2), _ptr(fw1), _ptr(fb1), _ptr(fw2), _ptr(fb2), _ptr(gate_ref),
                                   B, H, W, 32, st), "eg_se_gate_pre")
        L.check(lib.eg_se_gate_pre_planar(_ptr(y1p), _ptr(gap_p), tiles_p, _ptr(wp2), _ptr(s2), _ptr(t2), _ptr(fw1), _ptr(fb1), _ptr(fw2), _ptr(fb2),
                                          _ptr(gate_p), B, H, W, st), "eg_se_gate_pre_planar")
        assert float((gate_p - gate_ref).abs().max()) < 2e-6
    else:
        gate_p.fill_(0.7)
    # conv2 with the fused SE tail: relu(BN(conv(t1)) * gate + x), to planes and to fp32 NHWC
    out_ref = torch.empty(B, H, W, 32, device=d)
    L.check(lib.eg_conv3x3_se(_ptr(y1), _ptr(wp2), None, _ptr(s2), _ptr(t2), _ptr(gate_p), _ptr(x0), _ptr(out_ref), None, B, H, W, 32, 32, 1, 0, 0, pc, st),
            "eg_conv3x3_se")
    outp, outf = torch.empty(B * H * W * 32, device=d), torch.empty(B, H, W, 32, device=d)
    L.check(lib.eg_conv3x3_c32_planar(_ptr(y1p), _ptr(wp2), None, _ptr(s2), _ptr(t2), _ptr(gate_p), _ptr(xp), _ptr(outp), None, None, B, H, W, 0, pc, st),
            "eg_conv3x3_c32_planar (tail, planes)")
    L.check(lib.eg_conv3x3_c32_planar(_ptr(y1p), _ptr(wp2), None, _ptr(s2), _ptr(t2), _ptr(gate_p), _ptr(xp), None, _ptr(outf), None, B, H, W, 0, pc, st),
            "eg_conv3x3_c32_planar (tail, fp32)")
    assert rel_l2(outf.cpu().numpy(), out_ref.cpu().numpy()) < (2e-6 if prec == "bf16x3" else tol)   # same inputs, same arithmetic: fp32 output
    assert rel_l2(_from_planes(outp, B, H, W).cpu().numpy(), out_ref.cpu().numpy()) < tol
    # determinism: a second launch is bitwise identical
    outf2 = torch.empty_like(outf)
    L.check(lib.eg_conv3x3_c32_planar(_ptr(y1p), _ptr(wp2), None, _ptr(s2), _ptr(t2), _ptr(gate_p), _ptr(xp), None, _ptr(outf2), None, B, H, W, 0, pc, st),
            "eg_conv3x3_c32_planar")
    assert torch.equal(outf, outf2)

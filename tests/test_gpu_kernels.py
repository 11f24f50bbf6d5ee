"""Per-kernel parity: every block operator of include/emogest.h, called through the C ABI on the GPU, against
the CPU oracle / plain fp32 torch reference of the same op on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import build_mirror, rel_l2
from emotiongestures_amd.synth import hash_uniform

pytestmark = pytest.mark.gpu

# tolerance per arithmetic mode (relative L2 of the whole tensor)
TOL = {"f32": 2e-6, "bf16x3": 3e-5, "bf16": 2e-2}


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(key, shape, lo=-1.0, hi=1.0, seed=0):
    return torch.from_numpy(hash_uniform(key, shape, lo, hi, seed))


CONV_CASES = [   # cin, cout, stride, H, W, nchw
    (32, 32, 1, 24, 40, False), (32, 64, 2, 24, 44, False), (64, 64, 1, 16, 62, False), (64, 128, 2, 18, 62, False),
    (128, 128, 1, 12, 31, False), (128, 34, 1, 8, 31, True), (128, 60, 1, 8, 20, True), (32, 32, 1, 128, 124, False),
]


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("cin,cout,stride,H,W,nchw", CONV_CASES)
def test_conv3x3_epilogues(cin, cout, stride, H, W, nchw, prec):
    from emotiongestures_amd import ops
    B = 2
    x = T(f"x{cin}", (B, cin, H, W), -1, 1)
    w = T(f"w{cin}{cout}", (cout, cin, 3, 3), -0.1, 0.1)
    bias, scale, shift = T("b", (cout,), -0.2, 0.2), T("s", (cout,), 0.5, 1.5), T("t", (cout,), -0.3, 0.3)
    for relu in (True, False):
        ref = F.conv2d(x, w, bias, stride=stride, padding=1)
        if relu:
            ref = F.relu(ref)
        ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        xg = x.permute(0, 2, 3, 1).contiguous().to(dev())
        y, gap = ops.conv3x3(xg, w, bias, scale, shift, stride=stride, relu=relu, nchw_out=nchw, want_gap=True, precision=prec)
        got = y.cpu().view(ref.shape) if nchw else y.cpu().permute(0, 3, 1, 2)
        assert rel_l2(got.numpy(), ref.numpy()) < TOL[prec]
        # SE partial sums: sum over tiles == sum over pixels of the epilogue output
        assert rel_l2(gap.sum(1).cpu().numpy(), ref.sum((2, 3)).numpy()) < max(TOL[prec], 1e-5)


def test_conv3x3_rejects_bad_arguments():
    from emotiongestures_amd import ops
    from emotiongestures_amd._lib import EgError
    x = torch.zeros(1, 8, 8, 48, device=dev())
    with pytest.raises(EgError):
        ops.conv3x3(x, torch.zeros(48, 48, 3, 3))                # unsupported channel count
    with pytest.raises(EgError):
        ops.conv3x3(torch.zeros(1, 8, 8, 32), torch.zeros(32, 32, 3, 3))       # CPU tensor: no fallback
    with pytest.raises(EgError):
        ops.conv3x3(torch.zeros(1, 8, 8, 32, device=dev()), torch.zeros(32, 32, 3, 3), stride=3)


def test_stem_conv():
    from emotiongestures_amd import ops
    x = T("spec", (2, 128, 124), -80, 0)
    w, b = T("w", (32, 1, 3, 3), -0.3, 0.3), T("b", (32,), -0.1, 0.1)
    s, t = T("s", (32,), 0.5, 1.5), T("t", (32,), -0.2, 0.2)
    ref = F.relu(F.conv2d(x.unsqueeze(1), w, b, padding=1)) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    y = ops.stem_conv(x.to(dev()), w, b.to(dev()), s.to(dev()), t.to(dev()))
    assert rel_l2(y.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < 2e-6


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("layer,idx,H,W", [("layer1", 1, 32, 40), ("layer2", 0, 32, 44), ("layer2", 2, 16, 22), ("layer3", 0, 16, 22), ("layer3", 3, 8, 31)])
def test_se_basic_block_module(layer, idx, H, W, prec):
    """SEBasicBlock.forward (Full_model/ResNetBlocks.py:21-37) incl. the stride-2 downsample shortcut."""
    from oracle import emogest_oracle as O
    m = build_mirror("spatial", 34, 126, 4, 4)
    blk = getattr(m.audio_encoder.feat_extractor, layer)[idx]
    p = f"audio_encoder.feat_extractor.{layer}.{idx}"
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cin = blk.conv1.weight.shape[1]
    x = T("blk", (2, cin, H, W), -1, 1)
    ref = O.se_basic_block(sd, p, x, blk.stride)
    blk.to(dev()).eval()
    blk.precision = prec
    got = blk(x.to(dev())).cpu()
    assert got.shape == ref.shape
    assert rel_l2(got.numpy(), ref.numpy()) < TOL[prec] * 3


LIN_CASES = [(70, 512, 512), (68, 126, 128), (34, 2048, 512), (130, 300, 300), (5, 8, 64), (64, 512, 992)]


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("M,N,K", LIN_CASES)
def test_linear_epilogues(M, N, K, prec):
    from emotiongestures_amd import ops
    x, w, b = T("x", (M, K)), T("w", (N, K), -0.05, 0.05), T("b", (N,), -0.1, 0.1)
    r1, r2 = T("r1", (M, N)), T("r2", (M, N))
    d = dev()
    y = ops.linear(x.to(d), w, b, precision=prec).cpu()
    assert rel_l2(y.numpy(), F.linear(x, w, b).numpy()) < TOL[prec]
    y = ops.linear(x.to(d), w, b, res1=r1.to(d), relu=True, precision=prec).cpu()
    assert rel_l2(y.numpy(), F.relu(F.linear(x, w, b) + r1).numpy()) < TOL[prec]
    y = ops.linear(x.to(d), w, None, res1=r1.to(d), res2=r2.to(d), relu=True, precision=prec).cpu()
    assert rel_l2(y.numpy(), F.relu(F.relu(F.linear(x, w) + r1) + r2).numpy()) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_linear_causal_shift_matches_dilated_conv_tap(prec):
    """a_shift/a_seq: the x[t-d] tap of the causal dilated conv (Full_model/tcn.py:18-24)."""
    from emotiongestures_amd import ops
    B, Ln, Cc, dil = 3, 60, 300, 4
    x, w = T("x", (B * Ln, Cc)), T("w", (Cc, Cc), -0.05, 0.05)
    xs = torch.zeros(B, Ln, Cc)
    xs[:, dil:] = x.view(B, Ln, Cc)[:, :-dil]
    ref = F.linear(xs.view(B * Ln, Cc), w)
    y = ops.linear(x.to(dev()), w, a_shift=dil, a_seq=Ln, precision=prec).cpu()
    assert rel_l2(y.numpy(), ref.numpy()) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_linear_splitk(prec):
    from emotiongestures_amd import ops
    M, N, K = 6, 512, 34 * 512
    x, w, b = T("x", (M, K)), T("w", (N, K), -0.01, 0.01), T("b", (N,))
    y = ops.linear_splitk(x.to(dev()), w, b, relu=True, splits=34, precision=prec).cpu()
    assert rel_l2(y.numpy(), F.relu(F.linear(x, w, b)).numpy()) < TOL[prec] * 2


@pytest.mark.parametrize("rows,d", [(68, 512), (7, 128), (33, 2048), (1, 64)])
def test_layernorm(rows, d):
    from emotiongestures_amd import ops
    x, g, b = T("x", (rows, d), -3, 3), T("g", (d,), 0.5, 1.5), T("b", (d,), -0.5, 0.5)
    y = ops.layernorm(x.to(dev()), g.to(dev()), b.to(dev()), 1e-6).cpu()
    assert rel_l2(y.numpy(), F.layer_norm(x, (d,), g, b, 1e-6).numpy()) < 2e-6


@pytest.mark.parametrize("prec,tol", [("f32", 3e-6), ("bf16x3", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("B,H,Lq,Lk", [(2, 8, 34, 34), (1, 8, 60, 60), (2, 2, 120, 120), (3, 8, 34, 60), (1, 1, 1, 1), (1, 2, 70, 200), (2, 1, 17, 49)])
def test_attention(B, H, Lq, Lk, prec, tol):
    """ScaledDotProductAttention (Full_model/Modules.py:13-23) on MFMA in every arithmetic mode: ragged Lq != Lk, the 1x1 edge
    case, more than one 64-row query chunk, every key-tile instantiation (Lk <= 48 / 64 / 128 / 256)."""
    from emotiongestures_amd import ops
    D = H * 64
    q, k, v = T("q", (B, Lq, D), -2, 2), T("k", (B, Lk, D), -2, 2), T("v", (B, Lk, D))
    split = lambda t, L: t.view(B, L, H, 64).transpose(1, 2)
    attn = torch.softmax(torch.matmul(split(q, Lq) / 8.0, split(k, Lk).transpose(2, 3)), dim=-1)
    ref = torch.matmul(attn, split(v, Lk)).transpose(1, 2).reshape(B, Lq, D)
    out, a = ops.attention(q.to(dev()), k.to(dev()), v.to(dev()), H, want_attn=True, precision=prec)
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < tol
    assert rel_l2(a.cpu().numpy(), attn.numpy()) < tol
    out2 = ops.attention(q.to(dev()), k.to(dev()), v.to(dev()), H, precision=prec)         # without the probability output
    assert torch.equal(out2, out)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_mha_and_ffn_modules(prec):
    """MultiHeadAttention / PositionwiseFeedForward forward (Full_model/SubLayers.py:30-59,74-84) and a whole
    DecoderLayer (Full_model/Layers.py:50-58) at module level."""
    from oracle import emogest_oracle as O
    m = build_mirror("spatial", 34, 126, 4, 4)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = O.GenCfg()
    x, e = T("x", (2, 34, 512)), T("e", (2, 34, 512))
    layer = m.decoder.layer_stack[1].to(dev()).eval()
    layer.enc_attn.precision = layer.pos_ffn.precision = prec
    ref_a, ref_attn = O.multi_head_attention(sd, "decoder.layer_stack.1.enc_attn", x, e, e, cfg)
    ref = O.positionwise_ffn(sd, "decoder.layer_stack.1.pos_ffn", ref_a)
    got_a, got_attn = layer.enc_attn(x.to(dev()), e.to(dev()), e.to(dev()))
    assert rel_l2(got_a.cpu().numpy(), ref_a.numpy()) < TOL[prec] * 3
    assert rel_l2(got_attn.cpu().numpy(), ref_attn.numpy()) < TOL[prec] * 3
    out, _, _ = layer(x.to(dev()), e.to(dev()))
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < TOL[prec] * 3


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_tcn_module(prec):
    """TemporalConvNet.forward (Full_model/tcn.py:63): 3 levels, dilation 1/2/4, causal."""
    from oracle import emogest_oracle as O
    m = build_mirror("spatial", 34, 126, 4, 4)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = T("tcn", (2, 300, 60))
    ref = x
    cfg = O.GenCfg()
    for i in range(3):
        d, q = 2 ** i, f"text_encoder.tcn.network.{i}"
        out = ref
        for c in ("conv1", "conv2"):
            w = O.weight_norm_weight(sd[f"{q}.{c}.weight_v"], sd[f"{q}.{c}.weight_g"])
            out = F.relu(F.conv1d(out, w, sd[f"{q}.{c}.bias"], padding=d, dilation=d)[:, :, :60])
        ref = F.relu(out + ref)
    tcn = m.text_encoder.tcn.to(dev()).eval()
    tcn.precision = prec
    got = tcn(x.to(dev())).cpu()
    assert rel_l2(got.numpy(), ref.numpy()) < TOL[prec] * 3
    # causality: changing the last time step must not change earlier outputs
    x2 = x.clone()
    x2[:, :, -1] += 1.0
    got2 = tcn(x2.to(dev())).cpu()
    assert torch.equal(got2[:, :, :-1], got[:, :, :-1])


def test_reparameterize_and_add_rows():
    from emotiongestures_amd import ops
    mu, lv, eps = T("mu", (5, 32)), T("lv", (5, 32), -2, 2), T("eps", (5, 32), -3, 3)
    z = ops.reparameterize(mu.to(dev()), lv.to(dev()), eps.to(dev())).cpu()
    assert rel_l2(z.numpy(), (eps * torch.exp(0.5 * lv) + mu).numpy()) < 2e-6
    a, tab = T("a", (3, 34, 512)), T("tab", (34, 512))
    assert torch.equal(ops.add_rows(a.to(dev()), tab.to(dev()), period=34).cpu(), a + tab)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("M,N,K", [(136, 512, 512),        # 64 x 64 tiles (one workgroup per CU regime)
                                   (66500, 192, 96),       # 128 x 128 tiles (>= 1024 of them), odd tile counts on both axes, 3 K-steps
                                   (26112, 640, 1024)])    # 128 x 128 tiles, full ring turns
def test_linear_presplit_matches_fp64(prec, M, N, K):
    """eg_split_tiles + eg_linear_presplit (both tile sizes) vs a float64 product, with bias + residual + ReLU."""
    import ctypes as C

    from emotiongestures_amd import _lib as L
    from emotiongestures_amd import ops
    from emotiongestures_amd.engine import _ptr, _stream
    lib = L.load()
    x, w = T("px", (M, K)), T("pw", (N, K), -0.1, 0.1)
    bias, res = T("pb", (N,)), T("pr", (M, N))
    xd, rd, bd = x.to(dev()), res.to(dev()), bias.to(dev())
    wp, npad, kpad = ops.pack_linear_weight(w, dev())
    bp = torch.zeros(npad, device=dev()); bp[:N] = bd
    kp, mt = (K + 63) // 64 * 64, (M + 63) // 64
    img = torch.empty(2 * mt * 64 * kp, dtype=torch.int16, device=dev())
    L.check(lib.eg_split_tiles(_ptr(xd), K, M, K, _ptr(img), _stream(dev())), "eg_split_tiles")
    y = torch.empty(M, N, device=dev())
    L.check(lib.eg_linear_presplit(_ptr(img), K, _ptr(wp), kpad, _ptr(bp), _ptr(rd), None, N, _ptr(y), N, M, N, K, 1,
                                   L.precision_code(prec), _stream(dev())), "eg_linear_presplit")
    ref = torch.relu(x.double() @ w.double().T + bias.double() + res.double())
    assert rel_l2(y.cpu().numpy(), ref.numpy()) < TOL[prec]
    # same operands through eg_linear (in-kernel split): identical arithmetic, so identical bits
    y2 = torch.empty(M, N, device=dev())
    L.check(lib.eg_linear(_ptr(xd), K, _ptr(wp), kpad, _ptr(bp), _ptr(rd), None, N, _ptr(y2), N, M, N, K, 1, 0, 0,
                          L.precision_code(prec), _stream(dev())), "eg_linear")
    assert torch.equal(y, y2)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("M,N,K", [(2176, 512, 512), (2176, 2048, 512), (2176, 512, 2048),     # the headline's transformer products
                                   (200, 192, 96), (8390, 320, 1024), (64, 64, 32)])          # odd tile counts, a short K, a single tile
def test_linear_presplit_tile_variants_are_bitwise_equal(prec, M, N, K, monkeypatch):
    """The workgroup tile of the pre-split product is a scheduling choice: every variant (64 x 64, 128 x 64 with a 4- or 3-slot ring,
    128 x 128; forced through EG_GEMM_TILE, which the library reads per call) accumulates each output element over K in the same order with
    the same three split terms, so the results are bitwise identical -- to each other and to eg_linear's in-kernel split."""
    from emotiongestures_amd import _lib as L
    from emotiongestures_amd import ops
    from emotiongestures_amd.engine import _ptr, _stream
    lib = L.load()
    x, w = T("tx", (M, K)), T("tw", (N, K), -0.1, 0.1)
    bias, res = T("tb", (N,)), T("tr", (M, N))
    xd, rd = x.to(dev()), res.to(dev())
    wp, npad, kpad = ops.pack_linear_weight(w, dev())
    bp = torch.zeros(npad, device=dev()); bp[:N] = bias.to(dev())
    kp, mt = (K + 63) // 64 * 64, (M + 63) // 64
    img = torch.empty(2 * mt * 64 * kp, dtype=torch.int16, device=dev())
    L.check(lib.eg_split_tiles(_ptr(xd), K, M, K, _ptr(img), _stream(dev())), "eg_split_tiles")
    outs = {}
    for tile in ("64", "64r8", "128x64", "128x64r3", "128x64r6", "128", "128r4"):
        monkeypatch.setenv("EG_GEMM_TILE", tile)
        y = torch.full((M, N), float("nan"), device=dev())
        L.check(lib.eg_linear_presplit(_ptr(img), K, _ptr(wp), kpad, _ptr(bp), _ptr(rd), None, N, _ptr(y), N, M, N, K, 1,
                                       L.precision_code(prec), _stream(dev())), "eg_linear_presplit " + tile)
        torch.cuda.synchronize()
        outs[tile] = y
    monkeypatch.delenv("EG_GEMM_TILE")
    ref = torch.relu(x.double() @ w.double().T + bias.double() + res.double())
    assert rel_l2(outs["64"].cpu().numpy(), ref.numpy()) < TOL[prec]
    for tile in ("64r8", "128x64", "128x64r3", "128x64r6", "128", "128r4"):
        assert torch.equal(outs[tile], outs["64"]), tile


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("M,N,K", [(4352, 2048, 512), (4352, 512, 2048), (200, 320, 96), (130, 100, 36)])
def test_linear_wide_tile_of_the_in_kernel_split_product_is_bitwise_equal(prec, M, N, K, monkeypatch):
    """eg_linear on fp32 input (X split per consuming workgroup) has a 64 x 64 and a 64 x 128 workgroup tile (chosen by size; EG_GLDS_TILE forces
    one): same K order per output element, so the results are bitwise identical -- also with ragged M / N, odd tile counts and a short K."""
    from emotiongestures_amd import _lib as L
    from emotiongestures_amd import ops
    from emotiongestures_amd.engine import _ptr, _stream
    lib = L.load()
    x, w = T("gx", (M, K)), T("gw", (N, K), -0.1, 0.1)
    bias, res = T("gb", (N,)), T("gr", (M, N))
    xd, rd = x.to(dev()), res.to(dev())
    wp, npad, kpad = ops.pack_linear_weight(w, dev())
    bp = torch.zeros(npad, device=dev()); bp[:N] = bias.to(dev())
    outs = {}
    for tile in ("0", "1"):
        monkeypatch.setenv("EG_GLDS_TILE", tile)
        y = torch.full((M, N), float("nan"), device=dev())
        L.check(lib.eg_linear(_ptr(xd), K, _ptr(wp), kpad, _ptr(bp), _ptr(rd), None, N, _ptr(y), N, M, N, K, 1, 0, 0, L.precision_code(prec), _stream(dev())),
                "eg_linear tile " + tile)
        torch.cuda.synchronize()
        outs[tile] = y
    monkeypatch.delenv("EG_GLDS_TILE")
    ref = torch.relu(x.double() @ w.double().T + bias.double() + res.double())
    assert rel_l2(outs["0"].cpu().numpy(), ref.numpy()) < TOL[prec]
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_linear_random_shapes(prec):
    """eg_linear over 24 seeded shapes (ragged M / N, K not a multiple of the 32-deep step) vs float64, all epilogue options."""
    from emotiongestures_amd import ops
    rng = np.random.RandomState(1234)
    for case in range(24):
        M, N = int(rng.randint(1, 300)), int(rng.randint(1, 700))
        K = 4 * int(rng.randint(1, 280))
        relu = bool(case & 1)
        x, w = T(f"rx{case}", (M, K)), T(f"rw{case}", (N, K), -0.2, 0.2)
        b = T(f"rb{case}", (N,)) if case % 3 else None
        r1 = T(f"rr{case}", (M, N)) if case % 4 == 0 else None
        got = ops.linear(x.to(dev()), w, None if b is None else b.to(dev()), res1=None if r1 is None else r1.to(dev()), relu=relu,
                         precision=prec).cpu()
        ref = x.double() @ w.double().T
        if b is not None:
            ref = ref + b.double()
        if r1 is not None:
            ref = ref + r1.double()
        if relu:
            ref = torch.relu(ref)
        assert got.shape == (M, N)
        assert rel_l2(got.numpy(), ref.numpy()) < TOL[prec] * 2, (case, M, N, K)




@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_attention_mask_matches_reference_golden(prec):
    """MultiHeadAttention / ScaledDotProductAttention with the reference's mask argument (eg_attention_masked) against the REFERENCE's module
    (tests/golden/make_golden_mask.py): [B, 1, Lk] padding mask, [B, Lq, Lk] causal mask, fully masked rows; and mask = None still takes the
    fused block with the same result as an all-ones mask."""
    import os
    from emotiongestures_amd.modules import MultiHeadAttention
    from emotiongestures_amd.synth import hash_unit, load_synth_weights
    from test_oracle_golden import GOLDEN, _mask_inputs
    z = np.load(os.path.join(GOLDEN, "attention_mask.npz"))
    q, kv, masks = _mask_inputs()
    mha = load_synth_weights(MultiHeadAttention(8, 512, 64, 64, dropout=0.2), 31).eval().to(dev())
    mha.precision = prec
    qd, kd = torch.from_numpy(q).to(dev()), torch.from_numpy(kv).to(dev())
    tol = {"f32": 3e-6, "bf16x3": 3e-5}[prec]
    with torch.no_grad():
        for name, m in masks.items():
            y, attn = mha(qd, kd, kd, mask=torch.from_numpy(m).to(dev()))
            assert rel_l2(y.cpu().numpy()[:, :, ::4], z[f"{name}/out"]) < tol, name
            np.testing.assert_allclose(attn.cpu().numpy()[:, ::4], z[f"{name}/attn"], atol=10 * tol)
        y0, a0 = mha(qd, kd, kd)
        y1, a1 = mha(qd, kd, kd, mask=torch.ones(3, 1, 40, device=dev()))
        assert rel_l2(y1.cpu().numpy(), y0.cpu().numpy()) < tol and float((a1 - a0).abs().max()) < 10 * tol


def test_channel_split_convolutions_are_bitwise_the_unsplit_kernels(monkeypatch):
    """csrc/conv.hip conv3x3_bf16_kernel<..., SPLIT>: at small batches the 64 -> 64 and 128 -> 128 convolutions have fewer pixel tiles than the chip has
    CUs, so the launch function also spreads the output channels over workgroups (EG_CONV_SPLIT = 1 | 2 | 4 forces a split; default: by tile count).
    Same pixel -> lane map and K order: output map and pooling partials are bit-identical whatever the split -- so the result of a clip does not
    depend on the batch it travels in -- with the fused SE tail (gate + residual + ReLU) as well."""
    from emotiongestures_amd import _lib as L, ops
    from emotiongestures_amd.engine import _ptr, _stream
    dev = torch.device("cuda:0")
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    for (c, B, H, W, splits) in ((128, 1, 32, 31, ("2", "4")), (128, 3, 17, 40, ("2", "4")), (64, 1, 64, 62, ("2",)), (64, 2, 21, 33, ("2",)), (128, 16, 32, 31, ("2", "4"))):
        x = torch.randn(B, H, W, c, generator=g).to(dev)
        w = (torch.randn(c, c, 3, 3, generator=g) * 0.05)
        res = torch.randn(B, H, W, c, generator=g).to(dev)
        gate = torch.rand(B, c, generator=g).to(dev)
        sc, sh = (torch.rand(c, generator=g) + 0.5).to(dev), torch.randn(c, generator=g).to(dev)
        wp = ops.pack_conv3x3_weight(w, dev)[0]
        tiles = int(lib.eg_conv3x3_gap_tiles(H, W, c, c, 1))
        outs = {}
        for split in ("1",) + splits + (None,):
            if split is None:
                monkeypatch.delenv("EG_CONV_SPLIT", raising=False)
            else:
                monkeypatch.setenv("EG_CONV_SPLIT", split)
            y, gap = ops.conv3x3(x, w, relu=True, want_gap=True, precision="bf16x3", packed=(wp, None, None, None))
            y2, gap2 = torch.empty_like(x), torch.empty(B, tiles, c, device=dev)
            L.check(lib.eg_conv3x3_se(_ptr(x), _ptr(wp), None, _ptr(sc), _ptr(sh), _ptr(gate), _ptr(res), _ptr(y2), _ptr(gap2), B, H, W, c, c, 1, 0, 0, 2,
                                      _stream(dev)), "eg_conv3x3_se")
            outs[split] = (y.clone(), gap.clone(), y2, gap2)
        for split in splits + (None,):
            for a, b in zip(outs[split], outs["1"]):
                assert torch.equal(a, b), (c, B, H, W, split)


def test_one_clip_products_match_fp32_reference_and_the_tiled_kernels(monkeypatch):
    """csrc/gemm.hip gemm_skinny_kernel (M <= 64 rows; bf16x3: 16 output columns of a 64-row block per workgroup, K steps dealt to the four waves, operands straight from
    global memory, fixed-order fold): against a float64 reference within the split-bf16 bound, and against the tiled kernels (EG_GEMM_SKINNY=0) at
    summation-order noise -- ragged M / N / K, bias, ReLU, both residual forms and the causal row shift (a_shift) included."""
    from emotiongestures_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    for (M, K, N, kw) in ((34, 512, 512, {}), (1, 512, 1536, dict(bias=True)), (64, 2048, 512, dict(bias=True, res1=True)), (16, 300, 126, dict(bias=True, relu=True)),
                          (60, 128, 512, dict(bias=True, res1=True, relu=True, res2=True)), (34, 124, 512, dict(bias=True)), (47, 512, 2048, dict(bias=True, relu=True)),
                          (34, 512, 512, dict(bias=True, a_shift=4, a_seq=34)), (64, 512, 2048, dict(bias=True)), (50, 128, 126, dict(bias=True, a_shift=2, a_seq=50))):
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        b = torch.randn(N, generator=g).to(dev) if kw.get("bias") else None
        r1 = torch.randn(M, N, generator=g).to(dev) if kw.get("res1") else None
        r2 = torch.randn(M, N, generator=g).to(dev) if kw.get("res2") else None
        sh, sq = kw.get("a_shift", 0), kw.get("a_seq", 0)
        outs = {}
        for sk in ("0", "1"):
            monkeypatch.setenv("EG_GEMM_SKINNY", sk)
            outs[sk] = ops.linear(x, w, b, r1, r2, relu=bool(kw.get("relu")), a_shift=sh, a_seq=sq, precision="bf16x3")
        xs = x.double()
        if sh:
            xs = torch.zeros_like(xs)
            for m in range(M):
                if m % sq >= sh:
                    xs[m] = x[m - sh].double()
        ref = xs @ w.double().T
        if b is not None:
            ref = ref + b.double()
        if r1 is not None:
            ref = ref + r1.double()
        if kw.get("relu"):
            ref = ref.clamp_min(0)
        if r2 is not None:
            ref = (ref + r2.double()).clamp_min(0)
        refn = ref.cpu().numpy()
        for sk in ("0", "1"):
            e = rel_l2(outs[sk].double().cpu().numpy(), refn)
            assert e < 2e-5, (M, K, N, kw, sk, e)
        e = rel_l2(outs["1"].cpu().numpy(), outs["0"].cpu().numpy())
        assert e < 2e-6, (M, K, N, kw, e)

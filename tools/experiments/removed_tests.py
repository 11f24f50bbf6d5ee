"""Tests of the experiments in this directory as they stood when the kernels were in the product library (round 5).  Not collected by pytest
(not under tests/); they need the kernels pasted back as the README says."""

def test_conv32_teams_kernel_is_bitwise_the_persistent_kernel(monkeypatch):
    """csrc/conv.hip conv3x3_c32_teams_kernel (EG_CONV32_TEAMS = 2 | 3: teams of 4 waves per workgroup sharing ONE LDS copy of the 9 taps' weights,
    3 waves per SIMD) against the default persistent 32 -> 32 kernel: same arithmetic and summation order per output element, so identical bits --
    with the fused SE tail (gate + residual + ReLU in the epilogue), with the pooling partials, and on a tile count that leaves some teams idle."""
    from emotiongestures_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for (B, H, W) in ((3, 128, 124), (1, 20, 40), (5, 37, 33)):
        x = torch.randn(B, H, W, 32, generator=g).to(dev)
        w = (torch.randn(32, 32, 3, 3, generator=g) * 0.05)
        res = torch.randn(B, H, W, 32, generator=g).to(dev)
        wp = ops.pack_conv3x3_weight(w, dev)[0]
        outs = {}
        for teams in (None, "2", "3"):
            if teams is None:
                monkeypatch.delenv("EG_CONV32_TEAMS", raising=False)
            else:
                monkeypatch.setenv("EG_CONV32_TEAMS", teams)
            y, gap = ops.conv3x3(x, w, relu=True, want_gap=True, precision="bf16x3", packed=(wp, None, None, None))
            outs[teams] = (y.clone(), gap.clone())
        for teams in ("2", "3"):
            assert torch.equal(outs[teams][0], outs[None][0]) and torch.equal(outs[teams][1], outs[None][1]), (B, H, W, teams)
    # the whole generator (stage 1: six 32 -> 32 convolutions, three of them with the fused SE tail: gate + residual + ReLU in the epilogue)
    from conftest import build_mirror
    from emotiongestures_amd.synth import synth_inputs
    model = build_mirror("spatial", 34, 126, 4, 4, seed=4, precision="bf16x3").to(dev)
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synth_inputs(3, seed=4).items()}
    poses = {}
    for teams in (None, "3"):
        if teams is None:
            monkeypatch.delenv("EG_CONV32_TEAMS", raising=False)
        else:
            monkeypatch.setenv("EG_CONV32_TEAMS", teams)
        with torch.no_grad():
            poses[teams] = model(inp["spec"], inp["text"], inp["pre_pose"], inp["sampled"])[0].clone()
    assert torch.equal(poses["3"], poses[None])



def test_fused_ffn_slab_kernel_matches_the_two_launch_path(monkeypatch):
    """csrc/ffn.hip (opt-in: EG_FFN_FUSED): PositionwiseFeedForward as one slab kernel with the hidden in LDS.  One workgroup per slab ("1") is BITWISE
    the default two pre-split launches (same products per element in the same order, the hidden split by the same function); with the hidden split
    over 4 workgroups per slab ("4") the partial sums are folded by the LayerNorm in a fixed order: equal within 2e-5 at the pose, and repeatable."""
    model = build_mirror("spatial", 34, 126, 4, 4, seed=13, precision="bf16x3").to(dev())
    inp = synth_inputs(8, seed=13)
    g = {k: torch.from_numpy(v).to(dev()) for k, v in inp.items()}

    def run(mode):
        if mode is None:
            monkeypatch.delenv("EG_FFN_FUSED", raising=False)
        else:
            monkeypatch.setenv("EG_FFN_FUSED", mode)
        with torch.no_grad():
            return model(g["spec"], g["text"], g["pre_pose"], g["sampled"])[0].clone()
    ref, one, four, four_again = run(None), run("1"), run("4"), run("4")
    assert torch.equal(one, ref)
    assert torch.equal(four, four_again) and not torch.equal(four, ref)
    assert clip_rel_l2(four.cpu().numpy(), ref.cpu().numpy()) < 2e-5          # another summation order over the hidden chunks, six FFNs deep (measured 8e-6)

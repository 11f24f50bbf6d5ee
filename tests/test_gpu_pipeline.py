"""ClipPipeline: several independent batches in flight (own engine, workspace, hipGraph, stream per lane) must return, in
submission order, exactly what one eager forward per batch returns (the path is deterministic: bitwise equality)."""
import numpy as np
import pytest
import torch

from conftest import build_mirror
from emotiongestures_amd.synth import load_synth_weights, synth_audio, synth_inputs

pytestmark = pytest.mark.gpu


def _models(dev):
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.engine import MelFrontEnd
    gen = build_mirror("spatial", 34, 126, 4, 4, seed=3, precision="bf16x3").to(dev)
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 3).eval().to(dev)
    return gen, vae, MelFrontEnd(dev)


def _batch(B, seed, dev):
    inp = synth_inputs(B, 34, 126, 4, seed=seed)
    inp["audio"] = synth_audio(B, 64000, seed=seed)
    return {k: torch.from_numpy(inp[k]).to(dev) for k in ("audio", "text", "pre_pose", "label", "z")}


@pytest.mark.parametrize("lanes,branch", [(1, True), (3, False)])
def test_pipeline_matches_eager_in_order(lanes, branch):
    from emotiongestures_amd.pipeline import ClipPipeline
    dev = torch.device("cuda:0")
    B, n_batches = 4, 7
    batches = [_batch(B, 100 + i, dev) for i in range(n_batches)]
    pipe = ClipPipeline(_models(dev), batches[0], dev, lanes=lanes, branch_streams=branch)
    got = list(pipe.run(batches))
    assert len(got) == n_batches
    gen, vae, mel = _models(dev)
    for b, out in zip(batches, got):
        with torch.no_grad():
            ref = gen(mel(b["audio"], out_frames=124), b["text"], b["pre_pose"], vae.sample(b["label"], z=b["z"]))
        for a, r in zip(out, ref):
            assert torch.equal(a, r)
    # distinct batches really produce distinct poses (the comparison above is not vacuous)
    assert not torch.equal(got[0][0], got[1][0])
    with pytest.raises(ValueError):
        list(pipe.run([_batch(B + 1, 1, dev)]))
    # low-level driving as the bench does: replay lanes round-robin on resident inputs
    for _ in range(2 * lanes):
        lane = pipe.launch_next()
    pipe.synchronize()
    assert pipe.outputs(lane)[0].shape == (B, 34, 126)


def test_pipelines_of_both_lane_classes_share_one_generator():
    """a 1-lane (stand-alone tile policy, branches on side streams) and an N-lane pipeline (shared-chip tile policy) alive on ONE generator
    (bench.py's gpu_b1 leg beside the main pipeline).  The launch-policy hints are per pipeline, the generator keeps one engine per class: neither
    pipeline goes stale when the other is built or used, each arena is packed once, and both return what an eager forward returns."""
    from emotiongestures_amd.pipeline import ClipPipeline
    dev = torch.device("cuda:0")
    models = _models(dev)
    gen, vae, mel = models
    b4, b1 = _batch(4, 700, dev), _batch(1, 701, dev)
    many = ClipPipeline(models, b4, dev, lanes=2)
    one = ClipPipeline(models, b1, dev, lanes=1, branch_streams=True)
    assert gen.shared_chip is False and gen.concurrent is False           # the pipelines wrote nothing onto the shared module
    e_many, e_one = many._gen_engine(), one._gen_engine()
    assert e_many is not e_one and e_many.uploads == 1 and e_one.uploads == 1
    for _ in range(3):                                                     # interleaved use: no RuntimeError("... changed after capture")
        many.wait(many.launch_next())
        one.wait(one.launch_next())
    assert not many.stale() and not one.stale()
    many.refresh(); one.refresh()                                          # re-capture keeps each pipeline on its own class
    assert many._gen_engine() is e_many and one._gen_engine() is e_one and e_many.uploads == 1 and e_one.uploads == 1
    with torch.no_grad():
        ref4 = gen(mel(b4["audio"], out_frames=124), b4["text"], b4["pre_pose"], vae.sample(b4["label"], z=b4["z"]))
        ref1 = gen(mel(b1["audio"], out_frames=124), b1["text"], b1["pre_pose"], vae.sample(b1["label"], z=b1["z"]))
    got4, got1 = list(many.run([b4]))[0], list(one.run([b1]))[0]
    # the 128 x 128 product tile of the shared-chip class accumulates K in the same order as the 64 x 64 tile: same bits
    for a, r in zip(got4, ref4):
        assert torch.equal(a, r)
    for a, r in zip(got1, ref1):
        assert torch.equal(a, r)
    # a weight update reaches both classes: each repacks once, both pipelines notice
    with torch.no_grad():
        gen.post_projector[6].bias.add_(1.0)
    assert many.stale() and one.stale()
    got4b = list(many.run([b4]))[0]
    assert torch.allclose(got4b[0], got4[0] + 1.0, atol=1e-5) and many._gen_engine().uploads == 2


def test_pipeline_refuses_cpu():
    from emotiongestures_amd.pipeline import ClipPipeline
    with pytest.raises(RuntimeError):
        ClipPipeline((None, None, None), {}, "cpu")


def test_results_do_not_depend_on_concurrent_work():
    """A forward must return the same bits whether or not other kernels are resident on the GPU (lanes overlap a batch's
    transformer phase with another batch's convolutions).  Round 1 caught the attention kernel returning wrong lanes while
    an MFMA convolution was co-resident (tools/repro_pk_fma_under_mfma.py); this keeps the whole path under that load."""
    from emotiongestures_amd import _lib as L
    from emotiongestures_amd import ops
    from emotiongestures_amd.engine import _ptr, _stream
    dev = torch.device("cuda:0")
    lib = L.load()
    gen, vae, mel = _models(dev)
    gen.concurrent = False
    b = _batch(4, 300, dev)
    x = torch.randn(16, 32, 31, 128, device=dev)
    wp, _ = ops.pack_conv3x3_weight(torch.randn(128, 128, 3, 3) * 0.05, dev)
    y = torch.empty(16, 32, 31, 128, device=dev)
    pc = L.precision_code("bf16x3")

    def fwd():
        with torch.no_grad():
            return gen(mel(b["audio"], out_frames=124), b["text"], b["pre_pose"], vae.sample(b["label"], z=b["z"]))
    ref = [t.clone() for t in fwd()]
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for _ in range(12):
        with torch.cuda.stream(sB):
            for _ in range(150):
                lib.eg_conv3x3(_ptr(x), _ptr(wp), None, None, None, _ptr(y), None, 16, 32, 31, 128, 128, 1, 1, 0, pc, _stream(dev))
        with torch.cuda.stream(sA):
            out = fwd()
        torch.cuda.synchronize()
        for a, r in zip(out, ref):
            assert torch.equal(a, r)


def test_pipeline_streams_from_a_generator_without_host_syncs():
    """Batches come from a generator that drops every tensor right after yielding it, the consumer keeps only the pose and
    never synchronises the host in between: the allocator may then hand a dropped batch's block to the next batch while the
    lane's copy is still queued (record_stream / caller-stream clones).  Results must still match eager forwards."""
    from emotiongestures_amd.pipeline import ClipPipeline
    dev = torch.device("cuda:0")
    B, n_batches = 4, 24
    host = []
    for i in range(n_batches):
        inp = synth_inputs(B, 34, 126, 4, seed=500 + i)
        inp["audio"] = synth_audio(B, 64000, seed=500 + i)
        host.append({k: torch.from_numpy(inp[k]).pin_memory() for k in ("audio", "text", "pre_pose", "label", "z")})
    pipe = ClipPipeline(_models(dev), {k: v.to(dev) for k, v in host[0].items()}, dev, lanes=3)

    def produce():
        for h in host:
            yield {k: v.to(dev, non_blocking=True) for k, v in h.items()}      # dropped by the consumer loop immediately

    acc = torch.zeros(B, 34, 126, device=dev)
    poses = []
    for out in pipe.run(produce()):
        acc += out[0]                          # caller-stream work on the returned copies, no host sync
        poses.append(out[0])
    torch.cuda.synchronize()
    gen, vae, mel = _models(dev)
    want = torch.zeros_like(acc)
    for i, h in enumerate(host):
        b = {k: v.to(dev) for k, v in h.items()}
        with torch.no_grad():
            ref = gen(mel(b["audio"], out_frames=124), b["text"], b["pre_pose"], vae.sample(b["label"], z=b["z"]))[0]
        assert torch.equal(poses[i], ref), f"batch {i}"
        want += ref
    assert torch.equal(acc, want)


def test_pipeline_detects_a_stale_graph():
    """Captured graphs bake in pointers to the weight arena; after a weight update the low-level path must refuse to replay
    and run() must re-capture."""
    from emotiongestures_amd.pipeline import ClipPipeline
    dev = torch.device("cuda:0")
    gen, vae, mel = _models(dev)
    b = _batch(2, 900, dev)
    pipe = ClipPipeline((gen, vae, mel), b, dev, lanes=2)
    first = [t.clone() for t in next(iter(pipe.run([b])))]
    with torch.no_grad():
        gen.post_projector[6].bias.add_(0.25)          # bumps the parameter version -> new arena on the next engine() call
    assert pipe.stale()
    with pytest.raises(RuntimeError):
        pipe.launch_next()
    second = next(iter(pipe.run([b])))                 # re-captures
    torch.cuda.synchronize()
    assert torch.allclose(second[0], first[0] + 0.25, atol=1e-5)


def test_headline_batch_b64_matches_oracle():
    """BASELINE configs[1] at full size: B=64 TED clips from raw audio through ClipPipeline (mel -> CVAE sample -> generator,
    bf16x3, 2 lanes) against the CPU oracle on every clip.  Tolerance: per-clip relative L2 <= 1e-3 (north-star)."""
    from emotiongestures_amd.builders import clip_rel_l2
    from emotiongestures_amd.pipeline import ClipPipeline
    from oracle import emogest_oracle as O
    dev = torch.device("cuda:0")
    B = 64
    inp = synth_inputs(B, 34, 126, 4, seed=1000)
    inp["audio"] = synth_audio(B, 64000, seed=1000)
    gen, vae, mel = _models(dev)
    sd_g = {k: v.detach().cpu().clone() for k, v in gen.state_dict().items()}
    sd_v = {k: v.detach().cpu().clone() for k, v in vae.state_dict().items()}
    b = {k: torch.from_numpy(inp[k]).to(dev) for k in ("audio", "text", "pre_pose", "label", "z")}
    pipe = ClipPipeline((gen, vae, mel), b, dev, lanes=2)
    outs = list(pipe.run([b, b]))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0])
    t = {k: torch.from_numpy(v) for k, v in inp.items() if k != "audio"}
    with torch.no_grad():
        spec = torch.from_numpy(O.melspectrogram(inp["audio"], out_frames=124))
        ref = O.generator_forward(sd_g, O.GenCfg(), spec, t["text"], t["pre_pose"], O.cvae_sample(sd_v, t["label"], t["z"]))
    assert clip_rel_l2(outs[0][0].cpu().numpy(), ref[0].numpy()) < 1e-3
    assert float(np.abs(outs[0][3].cpu().numpy() - ref[3].numpy()).max()) < 1e-3          # emotion logits

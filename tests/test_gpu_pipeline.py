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


def test_pipeline_refuses_cpu():
    from emotiongestures_amd.pipeline import ClipPipeline
    with pytest.raises(RuntimeError):
        ClipPipeline((None, None, None), {}, "cpu")

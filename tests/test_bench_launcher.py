"""bench.py --gpus N launches itself (one worker process per GPU) when no outer launcher set WORLD_SIZE.
CPU side: the launcher plumbing (spawn, env, 127.0.0.1 rendezvous over gloo, barrier + max-reduce, ONE JSON line, exit codes)
with EG_BENCH_DRY=1, which skips every GPU call.  GPU side: a 1-GPU box must refuse --gpus 2 loudly."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env, timeout=300):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_workers_dry():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"EG_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                        # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True
    assert rec["max_elapsed"] == 2.0                        # MAX over ranks of (1 + rank)


def test_outer_launcher_env_rendezvous_dry():
    """What the driver's torchrun does: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, no EG_DIST_STORE -> env:// rendezvous."""
    procs = []
    for r in range(2):
        env = dict(os.environ, EG_BENCH_DRY="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29567")
        env.pop("EG_DIST_STORE", None)
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    recs = [json.loads(ln) for o in outs for ln in o[0].splitlines() if ln.startswith("{")]
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2 and recs[0]["max_elapsed"] == 2.0


def test_world_size_mismatch_is_refused():
    # an outer launcher that started 1 rank while --gpus says 2 (the round-1 bench silently reported n_gpus 1)
    env = dict(os.environ, EG_BENCH_DRY="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_failing_worker_fails_the_launcher():
    # a worker that cannot run (bad precision choice -> argparse exits 2) must surface as a non-zero parent status
    r = _run(["--gpus", "2", "--precision", "fp64"], {"EG_BENCH_DRY": "1"})
    assert r.returncode != 0


@pytest.mark.gpu
def test_more_gpus_than_visible_is_refused():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n)], {})
    assert r.returncode != 0 and "refusing to oversubscribe" in (r.stderr + r.stdout)

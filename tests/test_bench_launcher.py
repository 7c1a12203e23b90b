"""bench.py --gpus N launches its own ranks.  Here (no GPU): the same launcher / rendezvous / reducer / max-over-ranks timing code is
driven with 2 gloo ranks on CPU (`--mode ddp_selftest`: a toy torch module stands in for the model, nothing under oracle/ or the HIP library is used), and a
request for more GPUs than are visible must fail loudly instead of printing an n_gpus: 1 line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=e)


def test_launcher_spawns_two_gloo_ranks():
    r = _run("--gpus", "2", "--mode", "ddp_selftest", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["replicas_agree"] is True and d["buckets"] >= 2


def test_more_gpus_than_visible_fails_loudly():
    import torch
    n = torch.cuda.device_count()
    r = _run("--gpus", str(n + 2), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert f"{n + 2} GPUs requested, {n} visible" in r.stderr
    assert "n_gpus" not in r.stdout


def test_world_size_mismatch_is_an_error():
    r = _run("--gpus", "1", "--mode", "ddp_selftest", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr

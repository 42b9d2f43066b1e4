"""bench.py as its own launcher: `python bench.py --gpus N` with no WORLD_SIZE starts N ranks
itself, relays rank 0's JSON line and fails if any rank fails.  CPU part: argument / failure handling.  GPU part: the
whole thing once with two ranks on the one visible device (SMT_BENCH_ONE_DEVICE=1: gloo rendezvous, host-side
exchange) -- a rehearsal of the plumbing, the line says it is not a scaling measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_self_launch_relays_a_failing_rank():
    """No GPU here: both children die in torch.cuda.set_device; the parent must exit non-zero, promptly, with no
    JSON line.  (On a GPU box this test still holds: SMT_HIP_LIB points the children at a library that does not exist.)"""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--cpu-rows", "0"],
             {"SMT_BENCH_ONE_DEVICE": "1", "SMT_HIP_LIB": "/nonexistent/libsmt_hip.so"}, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_more_ranks_than_devices_is_refused_without_the_rehearsal_switch():
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    r = _run(["--gpus", str(n), "--steps", "1"])
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_device_rehearsal():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--pairs-per-step", "2"],
             {"SMT_BENCH_ONE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_joined"] == 2 and rec["config"]["world_size"] == 2
    assert rec["config"]["launcher"].startswith("self")
    assert rec["value"] > 0 and rec["scaling"] == "weak"
    s = rec["extra"]["cfg5_kitti_256pairs_strong"]
    assert s["pairs_total"] == 256 and s["pairs_this_rank"] == 128 and s["ranks"] == 2
    assert s["gathered_maps"] == [256, 375, 1242] and s["checksum_allreduce_equals_sum_of_gathered"]
    assert s["value_with_gather"] > 0 and s["value_without_gather"] >= s["value_with_gather"]

"""The real N > 1 path once: 2 ranks (fresh child processes, gloo rendezvous, both on cuda:0) run
shard.run_sharded(..., shard.adcensus_batch) on 5 KITTI-size pairs -- ragged shards 3 + 2 -- and the
gathered maps of both ranks must equal the oracle's (tests/golden/config_hashes.json, cfg 5 pairs 0-4).
Also the same callable under an initialised world_size-1 process group in this process."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "config_hashes.json")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_production_batch_and_gather(smt, O, tmp_path):
    rec = json.load(open(GOLD))["cfg5_kitti_d256_batch"]["pairs"]
    n_pairs, world = 5, 2
    port = str(_free_port())
    outs = [str(tmp_path / f"r{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_worker.py"), str(r), str(world), port,
                               str(n_pairs), outs[r]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log
    res = [json.load(open(o)) for o in outs]
    assert [r["shard"] for r in res] == [[0, 3], [3, 2]]
    for r in res:
        for b in range(n_pairs):
            assert r["left"][b] == rec[str(b)]["adcensus_disp_left"], (r["rank"], b)
            assert r["right"][b] == rec[str(b)]["adcensus_disp_right"], (r["rank"], b)
        assert abs(r["checksum"] - r["total"]) < 1e-6      # all-reduced shard sums == sum of the gathered maps


def test_world_size_1_process_group_runs_production_callable(smt, O):
    """run_sharded + adcensus_batch + gather_disparities + checksum under an initialised process group."""
    import torch.distributed as dist
    from stereo_match_traditional_amd import shard, synth
    rec = json.load(open(GOLD))["cfg5_kitti_d256_batch"]["pairs"]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        H, W, D, n = 375, 1242, 256, 2
        dev = torch.device("cuda:0")
        Ls, Rs = zip(*[synth.synth_pair(H, W, D, 1000 + b) for b in range(n)])
        L_all = torch.from_numpy(np.stack(Ls).astype(np.float32)).to(dev)
        R_all = torch.from_numpy(np.stack(Rs).astype(np.float32)).to(dev)

        def compute(L, R, Dd):
            dl, dr = shard.adcensus_batch(L, R, Dd)
            return dl.cpu(), dr.cpu()
        gl, gr = shard.run_sharded(L_all, R_all, D, compute)
        for b in range(n):
            assert "%016x" % O.fnv1a(gl[b].numpy()) == rec[str(b)]["adcensus_disp_left"]
            assert "%016x" % O.fnv1a(gr[b].numpy()) == rec[str(b)]["adcensus_disp_right"]
        assert abs(shard.checksum(gl) - float(gl.double().sum())) < 1e-6
    finally:
        dist.destroy_process_group()

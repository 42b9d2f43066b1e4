"""world_size-2 gloo tests of the N>1 path: pair sharding, ragged gather, checksum.
The compute callable is a CPU stand-in (pair-index-dependent maps) -- the data path of the
multi-GPU configuration has no collective, only this gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stereo_match_traditional_amd import shard


def test_shard_range_partitions_everything():
    for n in (0, 1, 5, 8, 255, 256, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = shard.shard_range(n, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(n))
            counts = [shard.shard_range(n, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_compute(L, R, D):
    # disparity "maps" that encode which pair they came from
    return L * 2.0 + 1.0, R * 3.0 + D


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W, D = 3, 5, 16
        base = torch.arange(n_pairs, dtype=torch.float32).reshape(n_pairs, 1, 1)
        L_all = base.expand(n_pairs, H, W).contiguous()
        R_all = (base + 100).expand(n_pairs, H, W).contiguous()
        gl, gr = shard.run_sharded(L_all, R_all, D, _fake_compute)
        ok = torch.equal(gl, L_all * 2.0 + 1.0) and torch.equal(gr, R_all * 3.0 + D)
        s, c = shard.shard_range(n_pairs, world, rank)
        chk = shard.checksum((L_all * 2.0 + 1.0)[s:s + c])
        ok = ok and abs(chk - float((L_all * 2.0 + 1.0).double().sum())) < 1e-6
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [4, 5, 1])
def test_two_rank_gather_gloo(n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}

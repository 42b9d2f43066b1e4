"""Child process of tests/test_shard_gpu.py: one rank of a world_size-N job (gloo rendezvous on
127.0.0.1, every rank on cuda:0) running the PRODUCTION sharded path -- shard.run_sharded with
shard.adcensus_batch, i.e. the C-ABI batch entry + the gather -- on KITTI-size synthetic pairs.
Writes the FNV-1a hashes of the gathered maps as JSON.  usage: _shard_worker.py rank world port pairs out.json"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, n_pairs, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereo_match_traditional_amd import shard, synth
    from oracle import oracle as O          # hashing only (test infrastructure)
    H, W, D = 375, 1242, 256
    dev = torch.device("cuda:0")
    Ls, Rs = zip(*[synth.synth_pair(H, W, D, 1000 + b) for b in range(n_pairs)])
    L_all = torch.from_numpy(np.stack(Ls).astype(np.float32)).to(dev)
    R_all = torch.from_numpy(np.stack(Rs).astype(np.float32)).to(dev)
    # gloo gathers host tensors: the shard's maps leave the GPU for the collective, as they would for
    # any consumer on the host
    def compute(L, R, Dd):
        dl, dr = shard.adcensus_batch(L, R, Dd)
        return dl.cpu(), dr.cpu()
    gl, gr = shard.run_sharded(L_all, R_all, D, compute)
    s, c = shard.shard_range(n_pairs, world, rank)
    chk = shard.checksum(gl[s:s + c])
    res = {"rank": rank, "shard": [s, c], "checksum": chk, "total": float(gl.double().sum()),
           "left": ["%016x" % O.fnv1a(gl[b].numpy()) for b in range(n_pairs)],
           "right": ["%016x" % O.fnv1a(gr[b].numpy()) for b in range(n_pairs)]}
    with open(out, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

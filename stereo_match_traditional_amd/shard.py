"""Image-pair sharding for the batched configuration (configs[4]: 256 KITTI-size pairs over
the 8 GPUs of a node).  Pairs are independent -- no halo, no exchange on the compute path --
so each rank runs a contiguous block of pairs on its own GPU; the only collective is the
final gather of the [pairs, H, W] disparity maps (1.86 MB each at 1242x375) plus an 8-byte
checksum all-reduce.  Works with any torch.distributed backend (RCCL/"nccl" on GPUs, gloo
on CPU for the tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_pairs, world, rank):
    """Contiguous block of pair indices for `rank`: (start, count); the first n_pairs % world
    ranks take one extra pair."""
    if n_pairs < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    q, r = divmod(n_pairs, world)
    count = q + (1 if rank < r else 0)
    start = rank * q + min(rank, r)
    return start, count


def gather_disparities(local, n_pairs, group=None):
    """all_gather the per-rank [count, H, W] maps into [n_pairs, H, W] in pair order (every
    rank gets the result).  Ranks may hold different counts (ragged shards): shards are
    padded to the largest count for the collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    counts = [shard_range(n_pairs, world, r)[1] for r in range(world)]
    cmax = max(counts) if counts else 0
    H, W = local.shape[-2:]
    pad = torch.zeros((cmax, H, W), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def checksum(t, group=None):
    """Sum of all disparities over all ranks (float64), the scalar cross-check of the gather."""
    s = t.to(torch.float64).sum().reshape(1)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(s, group=group)
    return float(s.item())


def run_sharded(L_all, R_all, D, compute, group=None):
    """Shard `n_pairs` pairs over the ranks, run `compute(L_shard, R_shard, D)` ->
    (dispL, dispR) on each, gather both maps.  `compute` is the HIP path in production
    (adcensus_batch below); tests substitute a CPU callable to exercise the N>1 plumbing."""
    n = L_all.shape[0]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    s, c = shard_range(n, world, rank)
    dl, dr = compute(L_all[s:s + c], R_all[s:s + c], D)
    return gather_disparities(dl, n, group), gather_disparities(dr, n, group)


def adcensus_batch(L, R, D, sigmaC=10.0, sigmaS=30.0):
    """The production `compute`: AD-Census both views + WTA for a [count, H, W] shard on this
    rank's GPU through the C ABI."""
    from .api import AD_Census
    c, H, W = L.shape
    dl = torch.empty((c, H, W), dtype=torch.float32, device=L.device)
    dr = torch.empty((c, H, W), dtype=torch.float32, device=L.device)
    if c == 0:
        return dl, dr
    adc = AD_Census().Initialize(L[0], R[0], D, H, W, sigmaC, sigmaS)
    adc.ComputeBatch(L.contiguous(), R.contiguous(), dl, dr)
    adc.status()
    adc.close()
    return dl, dr


def pipeline_batch(L8, R8, D, **params):
    """`compute` for run_sharded on configs[2]: the whole main.cpp pipeline (smt_pipeline_run_batch) for a
    [count, H, W] uint8 shard on this rank's GPU -> (LR-checked left maps, right maps)."""
    from .api import Pipeline
    c, H, W = L8.shape
    if c == 0:
        z = torch.empty((0, H, W), dtype=torch.float32, device=L8.device)
        return z, z.clone()
    pipe = Pipeline(H, W, D, L8.device, **params)
    dl, dr, _, _ = pipe.run(L8.contiguous(), R8.contiguous())
    pipe.status()
    pipe.close()
    return dl, dr

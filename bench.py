#!/usr/bin/env python3
"""Headline benchmark: AD-Census (both views' [H,W,D] cost volumes + WTA) at 1920x1080, D=192.

python bench.py --gpus N --steps K --warmup W
One process per GPU (torch.distributed over RCCL when N > 1); pairs are independent, so
each rank runs its own pairs with no data-path collective (weak scaling); one tiny
all_gather of the last disparity map + a checksum all_reduce stand for config 5's gather.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (H, W, D, seed)
    "adcensus_1080p_d192": (1080, 1920, 192, 3),   # BASELINE.json metric / configs[2] size
    "adcensus_720p_d128": (720, 1280, 128, 2),     # configs[1]
    "adcensus_kitti_d256": (375, 1242, 256, 1000), # configs[4] pair size
}


def cpu_baseline(H, W, D, seed, rows=256):
    """Oracle (CPU 'port' of AD-Census.h:271-380), 1 thread, on a band of `rows` rows of the
    SAME workload, both views + WTA.  Only this function touches oracle/."""
    from oracle import oracle as orc
    from stereo_match_traditional_amd import synth
    L, R = synth.synth_pair(H, W, D, seed)
    i0 = (H - rows) // 2
    i1 = i0 + rows
    t0 = time.perf_counter()
    vl = orc.adcensus_view(L, R, D, 10.0, 30.0, 0, i0, i1)
    vr = orc.adcensus_view(L, R, D, 10.0, 30.0, 1, i0, i1)
    orc.wta(vl[i0:i1])
    orc.wta(vr[i0:i1])
    dt = time.perf_counter() - t0
    hyp = rows * W * D
    base = {"value": round(hyp / dt / 1e6, 4), "unit": "Mdisp/s", "cores": 1, "kind": "port",
            "sample": f"rows {i0}..{i1 - 1} of the {W}x{H} D={D} pair ({hyp / 1e6:.1f} M hypotheses, "
                      f"both views + WTA, {dt:.1f} s, gcc -O2, 1 thread)"}
    # same loops, rows-parallel OpenMP build, all host cores (BASELINE.md plan, item 2) -- extra info
    try:
        import ctypes
        # a one-GPU box owns a 16-CPU share of the host whatever the affinity mask says
        ncores = min(len(os.sched_getaffinity(0)), 16)
        os.environ["OMP_NUM_THREADS"] = str(ncores)
        omp = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsmt_oracle_omp.so"))
        Lf = np.ascontiguousarray(L, np.float32)
        Rf = np.ascontiguousarray(R, np.float32)
        out = np.zeros((H, W, D), np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        # one untimed row first: starts the OpenMP thread pool and touches the pages
        omp.orc_adcensus_view(p(Lf), p(Rf), H, W, D, ctypes.c_float(10.0), ctypes.c_float(30.0), 0, i0, i0 + 1, p(out))
        t0 = time.perf_counter()
        for view in (0, 1):
            omp.orc_adcensus_view(p(Lf), p(Rf), H, W, D, ctypes.c_float(10.0), ctypes.c_float(30.0), view, i0, i1, p(out))
            d = np.empty((rows, W), np.float32)
            omp.orc_wta(p(out[i0:i1]), rows, W, D, p(d))
        dt2 = time.perf_counter() - t0
        base["all_cores"] = {"value": round(hyp / dt2 / 1e6, 3), "cores": ncores, "seconds": round(dt2, 2)}
    except OSError:
        pass
    return base


def pmc_traffic(workload):
    """HBM bytes per pair of the cost kernel (both launches) from the committed rocprofv3 PMC
    passes (profiles/pmc_traffic.json, written by tools/pmc_traffic.py from separate
    FETCH_SIZE / WRITE_SIZE runs with the gfx950 corrections of MI355X_MICROARCH.md).  PMC
    cannot be collected from inside an un-profiled run, so this is the last profiled value."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))[workload]
        return rec["hbm_bytes_per_pair"], rec["source"]
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="adcensus_1080p_d192", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs-per-step", type=int, default=1)
    ap.add_argument("--cpu-rows", type=int, default=256, help="rows in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # torch.distributed.run
    if world > 1 or launched:
        dist.init_process_group("nccl", device_id=dev)                   # RCCL on ROCm

    import stereo_match_traditional_amd as smt
    from stereo_match_traditional_amd import synth
    from stereo_match_traditional_amd._lib import lib
    lib()  # no fallback: fail here if the HIP library is missing

    H, W, D, seed = WORKLOADS[args.workload]
    P = args.pairs_per_step
    Ls, Rs = zip(*[synth.synth_pair(H, W, D, seed + 7919 * rank + b) for b in range(P)])
    Lb = torch.from_numpy(np.stack(Ls).astype(np.float32)).to(dev)
    Rb = torch.from_numpy(np.stack(Rs).astype(np.float32)).to(dev)
    dl = torch.empty((P, H, W), device=dev)
    dr = torch.empty((P, H, W), device=dev)
    adc = smt.AD_Census().Initialize(Lb[0], Rb[0], D, H, W, 10.0, 30.0)

    def step():
        adc.ComputeBatch(Lb, Rb, dl, dr)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events around the kernels of every 4th pair (each record costs ~3 us of stream time; all of
    # them when the run is short)
    stride = 4 if args.steps * P >= 40 else 1
    adc.timing(stride)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist.is_initialized():
        # config 5's only exchange: gather the disparity maps, all-reduce a checksum (shard.py)
        from stereo_match_traditional_amd import shard
        shard.gather_disparities(dl, world * P)
        shard.checksum(dl)
    barrier()
    dt = time.perf_counter() - t0
    prep_ms, cost_ms = adc.kernel_times()
    adc.timing(False)
    adc.status()

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        hyp_pair = H * W * D
        total_pairs = world * P * args.steps
        value = total_pairs * hyp_pair / dt / 1e6
        # dominant kernel = k_cost: writes both views' float32 volumes, 8 B per hypothesis
        alg_bytes = 8.0 * hyp_pair
        k_ms = float(np.mean(cost_ms)) if cost_ms else float("nan")
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload)
        out = {
            "metric": "Mdisparities/s (HxWxD/s) + ms/pair, AD-Census 1920x1080 D=192",
            "value": round(value, 2),
            "unit": "Mdisp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_pair": round(dt / (args.steps * P) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"AD-Census 9x7 both views + WTA, {W}x{H} D={D} ({args.workload})",
                       "pairs_per_step_per_gpu": P, "parallelism": f"pairs sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_cost (cost volume + fused WTA, both views)",
                         "kernel_ms": round(k_ms, 4), "tables_ms": round(float(np.mean(prep_ms)), 4),
                         "timed_launches": len(cost_ms), "timed_every": stride,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and args.cpu_rows > 0:
            out["cpu_baseline"] = cpu_baseline(H, W, D, seed, args.cpu_rows)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: AD-Census (both views' [H,W,D] cost volumes + WTA) at 1920x1080, D=192.

python bench.py --gpus N --steps K --warmup W
One process per GPU (torch.distributed over RCCL when N > 1); pairs are independent, so each rank runs
its own pairs with no data-path collective (weak scaling).  Config 5's only exchange -- one all_gather of
the disparity maps + a checksum all_reduce (shard.py) -- sits at the end of the timed region for the batched
workload (adcensus_kitti_d256; --gather forces it for any workload, --no-gather removes it); for the other
workloads it runs after the region.  Either way it is timed by itself (extra.gather_ms) and the rate with / without
it is reported.
A step is one smt_adcensus_compute_batch call over --pairs-per-step (default 8) resident pairs: the steady-state
throughput configuration (from the second pair of a batch on, the table kernels of the next pair overlap the
cost kernel of the current one); the latency of a lone pair is reported beside it (ms_single_pair_call).
`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts its own N ranks: the parent --
before anything touches the GPU -- spawns N children of this file with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would have set), relays rank 0's JSON line and exits
non-zero if any child does.  Under a launcher (WORLD_SIZE set) nothing is spawned; a WORLD_SIZE that differs from
--gpus is an error.  For every N the line also carries configs[4] as the strong-scaling case
(extra.cfg5_kitti_256pairs_strong: 256 pairs in total, 256/N per rank, the gather inside its timed region).
Prints ONE JSON line on rank 0.  After the timed region rank 0 also measures, outside `value`:
  roofline.sclk_mhz / store_ceiling_ms   in-kernel shader clock of the cost kernel and the same-run
                                         store-only ceiling of its store pattern (smt_adcensus_diag);
  extra.configs                          BASELINE.json configs 1-5 with per-stage HIP-event times;
  cpu_baseline                           the CPU oracle on a band of the same pair.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F64_VECTOR_TFLOPS = 78.6   # dense fp64 vector peak
F32_ADDS_PER_S = 78.6e12   # fp32 vector peak 157.3 TFLOP/s counts an FMA as 2: plain adds issue at half of it
SAD_U8_PER_S = 4 * 32 * 1024 * 2.4e9   # v_sad_u8: 4 byte differences + add per lane, 32 lanes per clock per SIMD, 1 024 SIMDs, 2.4 GHz
LDS_B32_TBS = 75.0         # ds_read_b32 aggregate rate with every CU streaming (MI355X_MICROARCH.md, LDS)

WORKLOADS = {
    # name: (H, W, D, seed, description)
    "adcensus_1080p_d192": (1080, 1920, 192, 3, "AD-Census 1920x1080 D=192"),   # BASELINE.json metric
    "adcensus_720p_d128": (720, 1280, 128, 2, "AD-Census 1280x720 D=128"),      # configs[1]
    "adcensus_kitti_d256": (375, 1242, 256, 1000, "AD-Census 1242x375 D=256"),  # configs[4] pair size
}


def cpu_baseline(H, W, D, seed, rows=256):
    """Oracle (CPU 'port' of AD-Census.h:271-380), 1 thread, on a band of `rows` rows of the
    SAME workload, both views + WTA.  Only this function touches oracle/."""
    from oracle import oracle as orc
    from stereo_match_traditional_amd import synth
    L, R = synth.synth_pair(H, W, D, seed)
    rows = min(rows, H)
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


def pmc_lds(key):
    """LDS counters of a profiled launch (profiles/pmc_lds.json <- tools/prof_pmc_scanline.sh)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_lds.json")))[key]
    except Exception:
        return None


def pmc_scanline():
    """HBM bytes per hypothesis of the three scanline kernels (profiles/pmc_scanline.json, same passes)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_scanline.json")))
    except Exception:
        return None


def smi_snapshot():
    """Best-effort rocm-smi readings (child process; never fatal): temperatures, clocks, power."""
    # under rocprofv3 the child would inherit the profiler's GPU-initialising preload and then exec rocm-smi's
    # interpreter, which the GPU boxes refuse: no snapshot there
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    try:
        r = subprocess.run(["rocm-smi", "--showtemp", "--showclocks", "--showpower", "--json"], capture_output=True,
                           text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(s in kl for s in ("temperature", "clock", "power")):
                keep[k] = v
        return keep
    except Exception:
        return None


def ev_timed(fn, reps, warm=1):
    """Mean milliseconds of fn() by HIP events on torch's current stream -- the stream api.py hands to
    every smt_* call, i.e. the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def extra_configs(dev, reps=5):
    """BASELINE.json configs 1-5 on this GPU, measured AFTER the headline's timed region (they are not
    part of `value`).  Times are HIP-event means; fractions are algorithmic bytes or flops
    over the spec peaks."""
    import stereo_match_traditional_amd as smt
    from stereo_match_traditional_amd import synth
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = {}

    # ---- config 1: SAD 5x5, 450x375, D=64 (left view) -------------------------------------------
    H, W, D = 375, 450, 64
    L, R = synth.synth_pair(H, W, D, 1)
    Lp, Rp = T(np.pad(L, 2, mode="edge")), T(np.pad(R, 2, mode="edge"))
    ms = ev_timed(lambda: smt.GetPointDepthLeft(Lp, Rp, D, 1), 20)
    taps = 25.0 * H * W * D                               # |a - b| + accumulate per window tap and hypothesis (Sad.h:15-20)
    out["cfg1_sad5x5_450x375_d64"] = {"ms_per_view": round(ms, 4), "Mdisp_s": round(H * W * D / ms / 1e3, 1),
                                      "bound": "valu-int (volume never stored)", "alg_byte_sads_per_view": taps,
                                      "frac_valu_sad_u8_peak": round(taps / (ms * 1e-3) / SAD_U8_PER_S, 4),
                                      "note": "10.8 M hypotheses in all (a 0.06 ms launch): window rows staged in LDS as byte-shifted dword "
                                              "copies, tap loop = LDS reads + v_sad_u8; 0.065 of the peak at 1080p D=128 9x9 (DESIGN.md 4)"}

    # ---- configs 2 and 5: AD-Census both views + WTA ---------------------------------------------
    for key, (H, W, D, seed, P) in {"cfg2_adcensus_720p_d128": (720, 1280, 128, 2, 8),
                                    "cfg5_adcensus_kitti_d256_batch": (375, 1242, 256, 1000, 16)}.items():
        Ls, Rs = zip(*[synth.synth_pair(H, W, D, seed + b) for b in range(P)])
        Lb, Rb = T(np.stack(Ls).astype(np.float32)), T(np.stack(Rs).astype(np.float32))
        dl, dr = torch.empty((P, H, W), device=dev), torch.empty((P, H, W), device=dev)
        adc = smt.AD_Census().Initialize(Lb[0], Rb[0], D, H, W, 10.0, 30.0)
        single = ev_timed(lambda: adc.ComputeBoth(dl[0], dr[0]), 40, warm=5)
        adc.timing(1)
        batch = ev_timed(lambda: adc.ComputeBatch(Lb, Rb, dl, dr), 10, warm=2) / P
        _, cost = adc.kernel_times()
        adc.timing(False)
        mhz, _, store_ms = adc.diag(20)
        adc.status()
        adc.close()
        V = H * W * D
        k_ms = float(np.mean(cost))
        out[key] = {"ms_per_pair_single": round(single, 4), "ms_per_pair_batched": round(batch, 4), "batch": P,
                    "Mdisp_s_batched": round(V / batch / 1e3, 1), "cost_kernel_ms": round(k_ms, 4),
                    "alg_bytes_per_pair": 8 * V, "frac_hbm_peak": round(8 * V / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "store_ceiling_ms": round(store_ms, 4), "sclk_mhz": round(mhz, 1)}
        del Lb, Rb, dl, dr

    # ---- config 3: the north-star pipeline, main.cpp:59-92 order ----------------------------------
    H, W, D = 1080, 1920, 192
    V = H * W * D
    L, R = synth.synth_pair(H, W, D, 3)
    Lf, Rf, Lu, Ru = T(L.astype(np.float32)), T(R.astype(np.float32)), T(L), T(R)
    adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0)
    dL, dR = torch.empty((H, W), device=dev), torch.empty((H, W), device=dev)
    st = {}
    st["adcensus"] = ev_timed(lambda: adc.ComputeBoth(dL, dR), reps)
    caL = smt.CrossArmAggregation().Initialize(H, W, 30, D, dev)
    caR = smt.CrossArmAggregation().Initialize(H, W, 30, D, dev)
    st["arms"] = ev_timed(lambda: (caL.ComputeArmLengths(Lu), caR.ComputeArmLengths(Ru)), reps)
    area = {}
    for nm, ca in (("left", caL), ("right", caR)):
        a = [m.double() for m in ca.arm_maps()]
        area[nm] = float(((a[0] + a[1] + 1) * (a[2] + a[3] + 1)).mean())
    aggL, aggR = torch.empty((H, W, D), device=dev), torch.empty((H, W, D), device=dev)
    # warm = 4: the first ~20 ms of a kernel on freshly allocated volumes run ~8 % slow (tools/agg_time.py, interleaved
    # rounds); the stage times are steady-state times
    st["aggregate_left"] = ev_timed(lambda: caL.AggregationVertical(adc.GetPtrLeft(), aggL, dL), reps, warm=4)
    st["aggregate_right"] = ev_timed(lambda: caR.AggregationVertical(adc.GetPtrRight(), aggR, dR), reps, warm=4)
    caL.status()
    caR.status()
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, dev)
    sout = torch.empty((H, W, D), device=dev)
    st["scanline"] = ev_timed(lambda: so.ScanLine(aggL, Lf, sout, dL), reps, warm=3)
    dLc = dL.clone()
    st["lrcheck"] = ev_timed(lambda: (dLc.copy_(dL), smt.LeftRightConsistency(W, H, 2, dLc, dR)), reps)
    total = sum(st.values())
    alg = {"adcensus": 8, "aggregate_left": 8, "aggregate_right": 8, "scanline": 44}
    stages = {}
    for k, ms in st.items():
        rec = {"ms": round(ms, 4)}
        if k in alg:
            rec["alg_bytes_per_hyp"] = alg[k]
            rec["frac_hbm_peak"] = round(alg[k] * V / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if k.startswith("aggregate"):
            # second roofline of the stage: the reference's add order is part of the contract, so the
            # in-order fp32 adds cannot be shared between pixels: sum(area) * D adds per view
            adds = area[k.split("_")[1]] * V
            rec["mean_rect_area"] = round(area[k.split("_")[1]], 2)
            rec["inorder_adds"] = adds
            rec["frac_f32_add_peak"] = round(adds / (ms * 1e-3) / F32_ADDS_PER_S, 4)
        if k == "scanline" and pmc_scanline():
            rec["hbm_traffic_profiled"] = pmc_scanline()
        stages[k] = rec
    out["cfg3_pipeline_1080p_d192"] = {
        "ms_per_pair": round(total, 4), "Mdisp_s": round(V / total / 1e3, 1), "alg_bytes_per_hyp": 68,
        "frac_hbm_peak": round(68 * V / (total * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "stages": stages}
    for o in (adc, caL, caR, so):
        o.close()
    del aggL, aggR, sout
    # the same pipeline through the one batched C-ABI entry (smt_pipeline_run_batch, 8 pairs per call):
    # launch gaps, staging and the device-side counts included, no host read-back
    pipe = smt.Pipeline(H, W, D, dev)
    Lb3, Rb3 = torch.stack([Lu] * 8), torch.stack([Ru] * 8)
    call_ms = ev_timed(lambda: pipe.run(Lb3, Rb3), 2) / 8
    pipe.status()
    pipe.close()
    out["cfg3_pipeline_1080p_d192"]["batched_entry_ms_per_pair"] = round(call_ms, 4)
    out["cfg3_pipeline_1080p_d192"]["batched_entry_frac_hbm_peak"] = round(68 * V / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    del Lb3, Rb3

    # ---- NCC (named in north_star): NCC_main.cpp's 21x21 window on a Middlebury-size pair ----
    H, W, win = 375, 450, 10
    for Dn in (64, 200):
        L, R = synth.synth_pair(H, W, 64, 1)
        Lt, Rt = T(L), T(R)
        ms = ev_timed(lambda: smt.NCC_algorithem(Lt, Rt, win, Dn), 3)
        rec = {"ms": round(ms, 4), "Mdisp_s": round((H - 2 * win) * (W - 2 * win) * Dn / ms / 1e3, 1),
               "bound": "lds (G + K - 1 ds_read_b32 per window row for the K slots of a lane + one broadcast read per v_dot4_u32_u8)"}
        lds = pmc_lds(f"ncc21x21_450x375_d{Dn}")
        if lds:
            # LDS wave-instructions of the k_ncc2 launch (rocprofv3 SQ_INSTS_LDS, committed) x 256 B each, over this run's time
            rec.update({"lds_wave_instructions": lds["SQ_INSTS_LDS"], "lds_bytes": lds["SQ_INSTS_LDS"] * 256.0,
                        "frac_lds_b32_peak": round(lds["SQ_INSTS_LDS"] * 256.0 / (ms * 1e-3) / (LDS_B32_TBS * 1e12), 4),
                        "lds_array_busy_frac_profiled": lds.get("lds_array_busy"), "lds_source": lds.get("source")})
        out[f"a23_ncc21x21_450x375_d{Dn}"] = rec

    # ---- CrossAggregator (named in north_star): 4 iterations x 2 passes at 1280x720, D=128 ----
    H, W, D = 720, 1280, 128
    L, R = synth.synth_pair(H, W, D, 2)
    bgr = T(np.repeat(L[..., None], 3, axis=2))
    cost0 = torch.rand((H, W, D), device=dev)
    ca = smt.CrossAggregator()
    ca.Initialize(W, H, 0, D, dev)
    ca.SetData(bgr, bgr, cost0)
    ca.SetParams(34, 17, 20, 6)
    ms = ev_timed(lambda: ca.Aggregate(4), 3)
    ca.close()
    out["a18_crossaggregator_720p_d128"] = {"ms_4_iterations": round(ms, 3), "alg_bytes_per_hyp": 64,
                                            "frac_hbm_peak": round(64.0 * H * W * D / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del bgr, cost0

    # ---- config 4: ASW 35x35 (winSize 16), 960x540, D=128, left view ------------------------------
    H, W, D, ws = 540, 960, 128, 16
    L, R = synth.synth_pair(H, W, D, 4)
    Lp, Rp = T(np.pad(L, ws + 1, mode="edge")), T(np.pad(R, ws + 1, mode="edge"))
    sp, cm = smt.asw_masks(ws, 50.0, 30.0, dev)
    ms = ev_timed(lambda: smt.AdaptiveSupportWeight(Lp, Rp, ws, D, sp, cm, 40), 2)
    flops = 8.0 * 35 * 35 * H * W * D
    out["cfg4_asw35x35_960x540_d128"] = {"ms_per_view": round(ms, 3), "Mdisp_s": round(H * W * D / ms / 1e3, 1),
                                         "alg_flops_per_view": flops, "TFLOP_s_f64": round(flops / ms / 1e9, 2),
                                         "frac_f64_vector_peak": round(flops / ms / 1e9 / F64_VECTOR_TFLOPS, 4)}
    return out


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher: be the launcher.  Nothing in this function touches the GPU
    (torch.cuda.device_count() does not initialise it on this image) and nothing is exec'ed: N plain child
    processes, one per GPU, with the rendezvous variables torch.distributed.run would have set."""
    import socket
    one_dev = os.environ.get("SMT_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < n and not one_dev:
        raise SystemExit(f"--gpus {n} but only {have} GPU(s) visible (SMT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 "
                         f"with a gloo rendezvous: a plumbing rehearsal, not a measurement)")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMT_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 owns stdout (the ONE JSON line); anything the other ranks print goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = time.time() + float(os.environ.get("SMT_BENCH_LAUNCH_TIMEOUT", "3000"))
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
        if rc != 0 or time.time() > deadline:
            # a dead rank leaves the others waiting in a collective: stop exactly the children started here
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(15)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    sys.exit(rc)


def cfg5_strong(dev, world, rank, on_host, total=256, reps=3):
    """BASELINE.json configs[4] as a strong-scaling case: 256 KITTI-size pairs (1242x375, D=256) in total,
    shard.shard_range(256, N, rank) of them on this rank in ONE smt_adcensus_compute_batch call, then the path's
    only exchange -- all_gather of the left maps + checksum all_reduce (shard.py) -- inside the timed region.
    Timed like the headline: barrier + synchronize on both sides, MAX over ranks.  Every rank calls this."""
    import torch.distributed as dist
    import stereo_match_traditional_amd as smt
    from stereo_match_traditional_amd import shard, synth
    H, W, D = 375, 1242, 256
    s, c = shard.shard_range(total, world, rank)
    Ls, Rs = zip(*[synth.synth_pair(H, W, D, 1000 + b) for b in range(s, s + c)])
    Lb = torch.from_numpy(np.stack(Ls).astype(np.float32)).to(dev)
    Rb = torch.from_numpy(np.stack(Rs).astype(np.float32)).to(dev)
    dl, dr = torch.empty((c, H, W), device=dev), torch.empty((c, H, W), device=dev)
    adc = smt.AD_Census().Initialize(Lb[0], Rb[0], D, H, W, 10.0, 30.0)
    multi = dist.is_initialized()

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def exchange():
        src = dl.cpu() if on_host else dl
        g = shard.gather_disparities(src, total)
        return g, shard.checksum(src)

    adc.ComputeBatch(Lb, Rb, dl, dr)                      # warm-up: kernels, RCCL channels of this shape
    g, chk = exchange()
    total_sum = float(g.double().sum())
    t_all, t_gather = [], []
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        adc.ComputeBatch(Lb, Rb, dl, dr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        exchange()
        barrier()
        t2 = time.perf_counter()
        t_all.append(t2 - t0)
        t_gather.append(t2 - t1)
    adc.status()
    adc.close()
    best = int(np.argmin(t_all))
    tt = torch.tensor([t_all[best], t_gather[best]], dtype=torch.float64, device="cpu" if on_host else dev)
    if multi:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t, tg = float(tt[0]), float(tt[1])
    hyp = total * H * W * D
    return {"pairs_total": total, "pairs_this_rank": c, "ranks": world, "scaling": "strong",
            "ms_total_with_gather": round(t * 1e3, 3), "gather_ms": round(tg * 1e3, 3),
            "value_with_gather": round(hyp / t / 1e6, 1), "value_without_gather": round(hyp / max(t - tg, 1e-9) / 1e6, 1),
            "unit": "Mdisp/s", "ms_per_pair_with_gather": round(t / total * 1e3, 4),
            "gathered_maps": list(g.shape), "checksum_allreduce_equals_sum_of_gathered": bool(abs(chk - total_sum) <= 1e-6 * max(1.0, abs(total_sum))),
            "timing": f"best of {reps} repetitions, max over ranks, barrier + synchronize on both sides"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="adcensus_1080p_d192", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs-per-step", type=int, default=8,
                    help="pairs per step PER GPU, one smt_adcensus_compute_batch call (config 5: 256/N); from the second pair "
                         "of a batch on, the table kernels of pair n+1 overlap the cost kernel of pair n")
    ap.add_argument("--cpu-rows", type=int, default=256, help="rows in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the disparity gather out of the timed region")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: put the disparity gather inside the timed region for any workload (default: only for "
                         "adcensus_kitti_d256, the batched configuration whose results are collected)")
    ap.add_argument("--no-extras", action="store_true", help="skip extra.configs (configs 1-5 after the timed region)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(sys.argv[1:], args.gpus)              # never returns; nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} is set but --gpus {args.gpus}: the launcher's world size and --gpus must agree "
                         f"(without a launcher, plain `python bench.py --gpus {args.gpus}` starts its own ranks)")
    import torch.distributed as dist
    # rehearsal mode for a one-GPU box: every rank on cuda:0, gloo rendezvous, collectives on host copies (RCCL
    # refuses two ranks on one device).  The line says so (config.backend); it is not a scaling measurement.
    one_dev = os.environ.get("SMT_BENCH_ONE_DEVICE") == "1" and world > 1
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # torch.distributed.run or self_launch
    backend = None
    if world > 1 or launched:
        backend = "gloo" if one_dev else "nccl"
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)               # RCCL on ROCm

    import stereo_match_traditional_amd as smt
    from stereo_match_traditional_amd import shard, synth
    from stereo_match_traditional_amd._lib import lib
    lib()  # no fallback: fail here if the HIP library is missing

    H, W, D, seed, desc = WORKLOADS[args.workload]
    P = args.pairs_per_step
    Ls, Rs = zip(*[synth.synth_pair(H, W, D, seed + 7919 * rank + b) for b in range(P)])
    Lb = torch.from_numpy(np.stack(Ls).astype(np.float32)).to(dev)
    Rb = torch.from_numpy(np.stack(Rs).astype(np.float32)).to(dev)
    dl = torch.empty((P, H, W), device=dev)
    dr = torch.empty((P, H, W), device=dev)
    adc = smt.AD_Census().Initialize(Lb[0], Rb[0], D, H, W, 10.0, 30.0)
    smi_before = smi_snapshot() if rank == 0 else None

    def step():
        adc.ComputeBatch(Lb, Rb, dl, dr)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def xdl():
        # what the exchange moves: the device maps over RCCL; host copies in the one-device gloo rehearsal
        return dl.cpu() if one_dev else dl

    # The pair axis shards with no data-path collective.  The batched configuration (configs[4]) ends with one exchange
    # -- all_gather of the disparity maps + a checksum all_reduce (shard.py) -- which belongs to its timed region; for
    # the other workloads the same exchange is run and timed after the region and reported beside `value`.
    gather_in_region = dist.is_initialized() and not args.no_gather and (args.gather or args.workload == "adcensus_kitti_d256")
    for _ in range(args.warmup):
        step()
    if dist.is_initialized() and not args.no_gather and args.warmup > 0:
        # part of the warm-up: the first collective of a shape sets up RCCL's channels (64 ms on one GPU)
        shard.gather_disparities(xdl(), world * P)
        shard.checksum(xdl())
    barrier()
    # HIP events around the kernels of ~32 pairs spread over the run (each record costs ~3 us of stream time and
    # a timed pair carries four)
    stride = max(1, (args.steps * P) // 32)               # ~32 timed pairs per run
    adc.timing(stride)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gather_ms = None

    def timed_gather():
        # the synchronize in front only separates the two clocks; a barrier would wait for it anyway
        torch.cuda.synchronize()
        tg = time.perf_counter()
        shard.gather_disparities(xdl(), world * P)
        shard.checksum(xdl())
        torch.cuda.synchronize()
        return (time.perf_counter() - tg) * 1e3

    if gather_in_region:
        gather_ms = timed_gather()
    barrier()
    dt = time.perf_counter() - t0
    if dist.is_initialized() and not args.no_gather and not gather_in_region:
        gather_ms = timed_gather()                        # outside the region: reported, not part of `value`
        barrier()
    prep_ms, cost_ms = adc.kernel_times()
    adc.timing(False)
    adc.status()

    tt = torch.tensor([dt, gather_ms or 0.0], dtype=torch.float64, device="cpu" if one_dev else dev)
    if dist.is_initialized():
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, gather_ms = float(tt[0].item()), (float(tt[1].item()) if gather_ms is not None else None)
    joined = 1
    if dist.is_initialized():
        ones = torch.ones(1, dtype=torch.float64, device="cpu" if one_dev else dev)
        dist.all_reduce(ones)                                # SUM over the communicator: how many ranks really took part
        joined = int(ones.item())

    # configs[4] as the strong-scaling case: every rank takes part (256 pairs in total, gather inside)
    strong = None
    if not args.no_extras:
        if world == 1:
            try:
                strong = cfg5_strong(dev, world, rank, one_dev)
            except Exception as e:                           # never lose the headline line to an extra
                strong = {"error": repr(e)}
        else:
            strong = cfg5_strong(dev, world, rank, one_dev)  # a rank that fails here ends the job: no silent hang

    # latency of a lone pair (one call, nothing to overlap with), outside the timed region
    single_ms = None
    if rank == 0:
        single_ms = ev_timed(lambda: adc.ComputeBatch(Lb[:1], Rb[:1], dl[:1], dr[:1]), 20, warm=2)
    # same-run diagnostics of the dominant kernel, right after the timed region (same DVFS / memory state)
    diag = None
    if rank == 0 and D % 64 == 0:
        mhz, stamped_ms, store_ms = adc.diag(20)
        diag = (mhz, stamped_ms, store_ms)

    if rank == 0:
        hyp_pair = H * W * D
        total_pairs = world * P * args.steps
        value = total_pairs * hyp_pair / dt / 1e6
        # dominant kernel = k_cost: writes both views' float32 volumes, 8 B per hypothesis
        alg_bytes = 8.0 * hyp_pair
        k_ms = float(np.mean(cost_ms)) if cost_ms else float("nan")
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload)
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_cost_fast2p / k_cost_fast2 (cost volume + fused WTA, both views; every launch of a batch but "
                          "the last also carries the next pair's table workgroups)",
                "kernel_ms": round(k_ms, 4), "tables_ms": round(float(np.mean(prep_ms)), 4),
                "timed_launches": len(cost_ms), "timed_every": stride,
                "algorithmic_bytes_per_launch": alg_bytes}
        if diag:
            mhz, stamped_ms, store_ms = diag
            roof.update({"sclk_mhz": round(mhz, 1), "store_ceiling_ms": round(store_ms, 4),
                         "store_ceiling_GBs": round(alg_bytes / (store_ms * 1e-3) / 1e9, 1),
                         "frac_of_store_ceiling": round(store_ms / k_ms, 4),
                         "stamped_kernel_ms": round(stamped_ms, 4),
                         "diag": "smt_adcensus_diag right after the timed region: store-only twin of the kernel "
                                 "(same grid, chunk order, streaming stores, same buffers) and the kernel with "
                                 "s_memtime/s_memrealtime stamps, 20 launches each"})
        tries, place_ms = adc.placement()
        roof["placement"] = {"candidate_pairs_tried": tries, "kept_pair_store_only_ms": round(place_ms, 4),
                             "note": "smt_adcensus_create keeps the fastest of up to 6 allocations of the two volumes "
                                     "(HBM write rate depends on the physical pages, DESIGN.md section 5)"}
        plain, nt_ms, plain_ms = adc.store_mode()
        roof["store_mode"] = {"chosen": "plain" if plain else "streaming (nt)", "calibration_nt_ms": round(nt_ms, 4),
                              "calibration_plain_ms": round(plain_ms, 4)}
        roof["smi_before"] = smi_before
        roof["smi_after"] = smi_snapshot()
        out = {
            "metric": f"Mdisparities/s (HxWxD/s) + ms/pair, {desc}",
            "value": round(value, 2),
            "unit": "Mdisp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_pair": round(dt / (args.steps * P) * 1e3, 4),
            "ms_single_pair_call": round(single_ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"AD-Census 9x7 both views + WTA, {W}x{H} D={D} ({args.workload})",
                       "pairs_per_step_per_gpu": P, "parallelism": f"pairs sharded over {world} GPU(s)",
                       "gather_in_timed_region": bool(gather_in_region),
                       "backend": ({"nccl": "nccl (RCCL)", "gloo": "gloo, every rank on cuda:0 (SMT_BENCH_ONE_DEVICE rehearsal, "
                                    "not a scaling measurement)"}[backend] if backend else "none (single process)"),
                       "ranks_joined": joined, "ranks_joined_how": "all_reduce(SUM) of 1 per rank over the job's communicator",
                       "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "launcher": ("self (bench.py spawned its ranks)" if os.environ.get("SMT_BENCH_SELF_LAUNCHED") == "1"
                                    else ("external (WORLD_SIZE was set)" if launched else "none"))},
            "roofline": roof,
        }
        extra = {}
        if gather_ms is not None:
            extra["gather_ms"] = round(gather_ms, 3)
            if gather_in_region:
                extra["value_without_gather"] = round(total_pairs * hyp_pair / (dt - gather_ms * 1e-3) / 1e6, 2)
            else:
                extra["value_with_gather"] = round(total_pairs * hyp_pair / (dt + gather_ms * 1e-3) / 1e6, 2)
        if strong is not None:
            extra["cfg5_kitti_256pairs_strong"] = strong
        if world == 1 and not args.no_extras:
            try:
                extra["configs"] = extra_configs(dev)
            except Exception as e:                       # never lose the headline line to an extra
                extra["configs_error"] = repr(e)
        if extra:
            out["extra"] = extra
        if world == 1 and args.cpu_rows > 0:
            out["cpu_baseline"] = cpu_baseline(H, W, D, seed, args.cpu_rows)
        print(json.dumps(out), flush=True)
    adc.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

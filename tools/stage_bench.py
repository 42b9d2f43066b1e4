#!/usr/bin/env python3
"""Per-stage timings of the full pipeline (configs[2]: AD-Census + CrossArm + 4-dir scanline +
LR check) and of the window matchers, with the algorithmic-bytes roofline fraction per stage.
Not the driver's bench (that is bench.py); this is the tuning harness."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereo_match_traditional_amd as smt  # noqa: E402
from stereo_match_traditional_amd import synth  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def pipeline(H, W, D, seed, noise, reps):
    L, R = synth.synth_pair(H, W, D, seed, noise)
    Lf = torch.from_numpy(L.astype(np.float32)).to(DEV)
    Rf = torch.from_numpy(R.astype(np.float32)).to(DEV)
    Lu, Ru = torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV)
    V = H * W * D
    res = {"H": H, "W": W, "D": D, "noise": noise}
    adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0)
    dL = torch.empty((H, W), device=DEV)
    dR = torch.empty((H, W), device=DEV)
    res["adcensus_ms"] = timed(lambda: adc.ComputeBoth(dL, dR), reps)
    caL = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    caR = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    res["arms_ms"] = timed(lambda: (caL.ComputeArmLengths(Lu), caR.ComputeArmLengths(Ru)), reps)
    caL.set_arm_walk(True)
    res["arms_walk_one_image_ms"] = timed(lambda: caL.ComputeArmLengths(Lu), reps)
    caL.set_arm_walk(False)
    res["arms_masks_one_image_ms"] = timed(lambda: caL.ComputeArmLengths(Lu), reps)
    aL, aR, aT, aB = [a.float() for a in caL.arm_maps()]
    area = (aL + aR + 1) * (aT + aB + 1)
    res["mean_arm"] = float((aL + aR + aT + aB).mean() / 4)
    res["mean_rect_area_left"] = float(area.mean())
    aggL = torch.empty((H, W, D), device=DEV)
    aggR = torch.empty((H, W, D), device=DEV)
    rL, rR, rT, rB = [a.float() for a in caR.arm_maps()]
    res["mean_rect_area_right"] = float(((rL + rR + 1) * (rT + rB + 1)).mean())
    for v in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
        caL.set_variant(v)
        res[f"aggregate_L_variant{v}_ms"] = timed(lambda: caL.AggregationVertical(adc.GetPtrLeft(), aggL), max(1, reps // 4))
    caL.set_variant(13)                                   # the default: 4x4 tiles, lock-step workgroups, scalar word
    res["aggregate_L_ms"] = timed(lambda: caL.AggregationVertical(adc.GetPtrLeft(), aggL), max(1, reps // 4))
    res["aggregate_R_ms"] = timed(lambda: caR.AggregationVertical(adc.GetPtrRight(), aggR, dR), max(1, reps // 4))
    caL.status()
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
    out = torch.empty((H, W, D), device=DEV)
    for k, name in enumerate(("left", "right", "up", "down")):
        res[f"scan_{name}_ms"] = timed(lambda: so.ScanPass(aggL, Lf, name), max(1, reps // 2))
    res["scanline_ms"] = timed(lambda: so.ScanLine(aggL, Lf, out, dL), max(1, reps // 2))
    res["lrcheck_ms"] = timed(lambda: smt.LeftRightConsistency(W, H, 2, dL.clone(), dR), reps)
    tot = res["adcensus_ms"] + res["arms_ms"] + res["aggregate_L_ms"] + res["aggregate_R_ms"] + res["scanline_ms"] + res["lrcheck_ms"]
    res["pipeline_ms"] = tot
    res["pipeline_Mdisp_s"] = V / tot / 1e3
    res["alg_GBs"] = {"adcensus": 8 * V / res["adcensus_ms"] / 1e6, "scanline(44B)": 44 * V / res["scanline_ms"] / 1e6,
                      "aggregate_L(8B)": 8 * V / res["aggregate_L_ms"] / 1e6}
    return res


def matchers(reps):
    res = {}
    # config 1: SAD 5x5 450x375 D=64
    H, W, D = 375, 450, 64
    L, R = synth.synth_pair(H, W, D, 1)
    Lp = torch.from_numpy(np.pad(L, 2, mode="edge")).to(DEV)
    Rp = torch.from_numpy(np.pad(R, 2, mode="edge")).to(DEV)
    res["sad_cfg1_left_ms"] = timed(lambda: smt.GetPointDepthLeft(Lp, Rp, D, 1), reps)
    # config 4: ASW 35x35 960x540 D=128 (winSize=16)
    H, W, D, ws = 540, 960, 128, 16
    L, R = synth.synth_pair(H, W, D, 4)
    Lp = torch.from_numpy(np.pad(L, ws + 1, mode="edge")).to(DEV)
    Rp = torch.from_numpy(np.pad(R, ws + 1, mode="edge")).to(DEV)
    sp, cm = smt.asw_masks(ws, 50.0, 30.0, DEV)
    res["asw_cfg4_left_ms"] = timed(lambda: smt.AdaptiveSupportWeight(Lp, Rp, ws, D, sp, cm, 40), 1)
    res["asw_cfg4_TFLOPs_f64"] = 8 * 35 * 35 * H * W * D / res["asw_cfg4_left_ms"] / 1e9
    # CrossAggregator (a18) at 720p D=128: arms + 4 iterations x 2 passes
    H, W, D = 720, 1280, 128
    L, R = synth.synth_pair(H, W, D, 2)
    bgr = torch.from_numpy(np.repeat(L[..., None], 3, axis=2).copy()).to(DEV)
    cost = torch.rand((H, W, D), device=DEV)
    ca = smt.CrossAggregator()
    ca.Initialize(W, H, 0, D, DEV)
    ca.SetData(bgr, bgr, cost)
    ca.SetParams(34, 17, 20, 6)
    res["crossagg_720p_d128_4iters_ms"] = timed(lambda: ca.Aggregate(4), 3)
    res["crossagg_alg_GBs(8B x 8 passes)"] = 64 * H * W * D / res["crossagg_720p_d128_4iters_ms"] / 1e6
    ca.close()
    # NCC 21x21 450x375 D=64
    H, W, D = 375, 450, 64
    L, R = synth.synth_pair(H, W, D, 1)
    res["ncc_21x21_450x375_d64_ms"] = timed(lambda: smt.NCC_algorithem(torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV), 10, D), 1)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1080p")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--matchers", action="store_true")
    a = ap.parse_args()
    H, W, D, seed = {"1080p": (1080, 1920, 192, 3), "720p": (720, 1280, 128, 2), "small": (375, 450, 64, 1)}[a.size]
    print(json.dumps(pipeline(H, W, D, seed, a.noise, a.reps), indent=1), flush=True)
    if a.matchers:
        print(json.dumps(matchers(a.reps), indent=1), flush=True)

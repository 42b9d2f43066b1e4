#!/usr/bin/env python3
"""Generates tests/golden/config_hashes.json: FNV-1a hashes (oracle/smt_oracle.c:orc_fnv1a) of every
volume and map of BASELINE.json configs 2, 3 and 5 at FULL size, computed by the CPU oracle in the build
container on the SURVEY 8(d) synthetic pairs, one stage at a time.  The `-m gpu` tests hash the device
results of the HIP path and compare (tests/test_config_hashes_gpu.py), so that the up/down scanline
passes, the right-view aggregation and the LR check are compared at 1920x1080x192 and not only
property-checked.

The oracle is PARITY UNPINNED for these stages (no reference build without OpenCV, no reference
fixtures); these hashes pin the HIP path to the oracle, not the oracle to the reference.

Run:  SMT_ORACLE_OMP=1 python tests/golden/make_config_hashes.py [cfg2] [cfg3] [cfg5] [a18] [--pairs N]
(OpenMP build of the same oracle file: planes / rows / lines are independent, results identical.)
cfg3 needs ~12 GB of host memory and ~10 min on 8 cores; cfg5 (256 pairs) ~35 min.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ.setdefault("SMT_ORACLE_OMP", "1")
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(HERE, "config_hashes.json")


def hx(a):
    return "%016x" % O.fnv1a(a)


def load():
    return json.load(open(OUT)) if os.path.exists(OUT) else {}


def save(db):
    with open(OUT, "w") as f:
        json.dump(db, f, indent=1, sort_keys=True)
        f.write("\n")


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def adcensus_stage(L, R, D, rec):
    vl = O.adcensus_view(L, R, D, 10.0, 30.0, 0)
    rec["adcensus_vol_left"] = hx(vl)
    dl = O.wta(vl)
    rec["adcensus_disp_left"] = hx(dl)
    vr = O.adcensus_view(L, R, D, 10.0, 30.0, 1)
    rec["adcensus_vol_right"] = hx(vr)
    dr = O.wta(vr)
    rec["adcensus_disp_right"] = hx(dr)
    return vl, vr


def cfg2(db):
    H, W, D, seed = 720, 1280, 128, 2
    L, R = O.synth_pair(H, W, D, seed)
    rec = {"H": H, "W": W, "D": D, "seed": seed, "sigmaC": 10.0, "sigmaS": 30.0}
    adcensus_stage(L, R, D, rec)
    db["cfg2_adcensus_720p_d128"] = rec
    save(db)
    log("cfg2 done")


def cfg3(db):
    H, W, D, seed = 1080, 1920, 192, 3
    L, R = O.synth_pair(H, W, D, seed)
    rec = {"H": H, "W": W, "D": D, "seed": seed, "sigmaC": 10.0, "sigmaS": 30.0, "tau": 30, "p1": 10, "p2": 150,
           "gate": 2}
    vl, vr = adcensus_stage(L, R, D, rec)
    log("cfg3 adcensus done")
    # main.cpp:67-84: Initialize(tao=30) + four arm passes + AggregationVertical + WTA, per view
    armsL = O.arms_all(L)
    armsR = O.arms_all(R)
    for nm, a in zip(("left", "right", "top", "bottom"), armsL):
        rec["arms_leftimg_" + nm] = hx(a)
    for nm, a in zip(("left", "right", "top", "bottom"), armsR):
        rec["arms_rightimg_" + nm] = hx(a)
    area = lambda a: float(((a[0] + a[1] + 1).astype(np.int64) * (a[2] + a[3] + 1)).mean())
    rec["mean_rect_area_left"] = area(armsL)
    rec["mean_rect_area_right"] = area(armsR)
    aggL, oob = O.aggregate_rect(vl, armsL, 0)
    assert oob == 0
    del vl
    rec["agg_vol_left"] = hx(aggL)
    rec["agg_disp_left"] = hx(O.wta(aggL))
    log("cfg3 aggregation left done")
    aggR, oob = O.aggregate_rect(vr, armsR, 0)
    assert oob == 0
    del vr
    rec["agg_vol_right"] = hx(aggR)
    dR = O.wta(aggR)
    rec["agg_disp_right"] = hx(dR)
    del aggR
    log("cfg3 aggregation right done")
    save({**db, "cfg3_pipeline_1080p_d192": rec})
    # main.cpp:86-89: ScanlineOptimizer on the LEFT aggregated volume, guided by the float gray left image
    gray = L.astype(np.float32)
    total = None
    for which in ("left", "right", "up", "down"):
        pv = O.scan_pass(aggL, gray, 10, 150, which)
        rec["scan_path_" + which] = hx(pv)
        total = pv if total is None else total + pv       # ((left+right)+up)+down, ScanlineOptimizer.h:124
        del pv
        log("cfg3 scan", which, "done")
    rec["scan_sum"] = hx(total)
    dL = O.wta(total)
    rec["scan_disp"] = hx(dL)
    del total, aggL
    # main.cpp:92: LeftRightConsistency(col, row, gate=2, leftDisp, rightDisp, ...)
    out, cls, nocc, nmis = O.lrcheck(dL, dR, 2)
    rec["lr_disp"] = hx(out)
    rec["lr_cls"] = hx(cls)
    rec["lr_n_occlusion"] = int(nocc)
    rec["lr_n_mismatch"] = int(nmis)
    db["cfg3_pipeline_1080p_d192"] = rec
    save(db)
    log("cfg3 done")


def cfg5(db, pairs):
    H, W, D = 375, 1242, 256
    rec = db.get("cfg5_kitti_d256_batch", {"H": H, "W": W, "D": D, "seed0": 1000, "sigmaC": 10.0, "sigmaS": 30.0,
                                           "pairs": {}})
    for b in range(pairs):
        if str(b) in rec["pairs"]:
            continue
        L, R = O.synth_pair(H, W, D, 1000 + b)
        r = {}
        adcensus_stage(L, R, D, r)
        rec["pairs"][str(b)] = r
        if b % 8 == 7 or b == pairs - 1:
            db["cfg5_kitti_d256_batch"] = rec
            save(db)
            log("cfg5 pair", b, "done")
    db["cfg5_kitti_d256_batch"] = rec
    save(db)


def a18(db):
    """CrossAggregator at its benchmark size, by the REFERENCE's own code (oracle/_ref, built from
    /root/reference/CBLSM/cross_aggregator.cpp by oracle/Makefile) -- the one pinned stage.  cost_init = config 2's
    left AD-Census volume, BGR = gray + (LCG byte mod 3) per channel, ADCensusOption's parameters."""
    from stereo_match_traditional_amd import synth
    H, W, D, seed = 720, 1280, 128, 2
    L, R = O.synth_pair(H, W, D, seed)
    bgr = np.clip(L.astype(np.int32)[..., None] + (synth.lcg_bytes(seed + 500, H * W * 3)[0] % 3).reshape(H, W, 3), 0, 255).astype(np.uint8)
    cost = O.adcensus_view(L, R, D, 10.0, 30.0, 0)
    rec = {"H": H, "W": W, "D": D, "seed": seed, "bgr_seed": seed + 500, "L1": 34, "L2": 17, "t1": 20, "t2": 6, "iters": 4,
           "cost_init": "adcensus_vol_left of cfg2", "cost_init_hash": hx(cost)}
    assert O.have_ref(), "oracle/_ref missing: make -C oracle ref (needs /root/reference)"
    t = time.time()
    arms, out = O.ref_crossagg(bgr, cost, 34, 17, 20, 6, 4)
    log("a18 reference build done in %.0f s" % (time.time() - t))
    arms2, out2 = O.crossagg(bgr, cost, 34, 17, 20, 6, 4)
    assert np.array_equal(arms, arms2) and np.array_equal(out.view(np.uint32), out2.view(np.uint32)), "oracle != reference build"
    rec["arms"] = hx(arms)
    rec["cost"] = hx(out)
    rec["disp"] = hx(O.wta(out))
    rec["produced_by"] = "reference build (oracle/_ref/libcrossagg_ref.so); the oracle restatement gives the same bits"
    db["a18_crossaggregator_720p_d128"] = rec
    save(db)
    log("a18 done")


if __name__ == "__main__":
    O.build()
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pairs = 256
    if "--pairs" in sys.argv:
        pairs = int(sys.argv[sys.argv.index("--pairs") + 1])
        args = [a for a in args if a != str(pairs)]
    which = args or ["cfg2", "cfg3", "cfg5"]
    db = load()
    if "cfg2" in which:
        cfg2(db)
    if "cfg3" in which:
        cfg3(db)
    if "cfg5" in which:
        cfg5(db, pairs)
    if "a18" in which:
        a18(db)

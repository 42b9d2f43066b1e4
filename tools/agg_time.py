#!/usr/bin/env python3
"""Times the rectangle aggregation of both views at 1920x1080 D=192 (HIP events, 5 reps) for the library named by
SMT_HIP_LIB -- used to A/B kernel builds (knock-outs, variants) in separate processes on one box.
usage: python tools/agg_time.py [tag]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth

DEV = torch.device("cuda:0")
H, W, D = 1080, 1920, 192
L, R = synth.synth_pair(H, W, D, 3)
Lu, Ru = torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV)
adc = smt.AD_Census().Initialize(Lu.float(), Ru.float(), D, H, W, 10.0, 30.0, placement_search=False, store_calibration=False)
adc.ComputeBoth()
out = torch.empty((H, W, D), device=DEV)
res = {"tag": sys.argv[1] if len(sys.argv) > 1 else "", "lib": os.environ.get("SMT_HIP_LIB", "default"),
       "SMT_AGG_WAVES": os.environ.get("SMT_AGG_WAVES")}
variants = [int(v) for v in os.environ.get("AGG_VARIANTS", "13").split(",")]
for name, img, vol in (("left", Lu, adc.GetPtrLeft()), ("right", Ru, adc.GetPtrRight())):
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(img)
    # interleaved rounds: the first measurement of a process runs ~8 % slower than later ones whatever the variant
    # (clock / memory state), so every variant is timed in every round and the rounds are reported one by one
    rounds = int(os.environ.get("AGG_ROUNDS", "3"))
    for rnd in range(rounds):
        for v in variants:
            ca.set_variant(v)
            for _ in range(2):
                ca.AggregationVertical(vol, out)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                ca.AggregationVertical(vol, out)
            b.record()
            torch.cuda.synchronize()
            res.setdefault(f"{name}_v{v}_ms", []).append(round(a.elapsed_time(b) / 5, 4))
    ca.close()
print(json.dumps(res), flush=True)

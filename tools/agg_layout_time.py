#!/usr/bin/env python3
"""Workgroup -> image mapping of the default aggregation kernel (variant 13) at 1920x1080 D=192: strip width and sweep
order (smt_crossarm_set_strip_width / set_sweep), interleaved rounds, both views.
usage: python tools/agg_layout_time.py [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth

DEV = torch.device("cuda:0")
H, W, D = 1080, 1920, 192
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L, R = synth.synth_pair(H, W, D, 3)
Lu, Ru = torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV)
adc = smt.AD_Census().Initialize(Lu.float(), Ru.float(), D, H, W, 10.0, 30.0, placement_search=False, store_calibration=False)
adc.ComputeBoth()
out = torch.empty((H, W, D), device=DEV)
res = {}
configs = [(8, 0), (8, 1), (16, 0), (16, 1), (32, 0), (32, 1), (64, 1)]
for name, img, vol in (("left", Lu, adc.GetPtrLeft()), ("right", Ru, adc.GetPtrRight())):
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(img)
    ref = torch.empty((H, W, D), device=DEV)
    ca.AggregationVertical(vol, ref)
    for rnd in range(rounds):
        for sw, sweep in configs:
            ca.set_variant(13); ca.set_strip_width(sw); ca.set_sweep(sweep)
            for _ in range(2):
                ca.AggregationVertical(vol, out)
            torch.cuda.synchronize()
            if rnd == 0:
                assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (sw, sweep)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                ca.AggregationVertical(vol, out)
            b.record()
            torch.cuda.synchronize()
            res.setdefault(f"{name}_sw{sw}_sweep{sweep}_ms", []).append(round(a.elapsed_time(b) / 5, 4))
    ca.close()
print(json.dumps(res), flush=True)

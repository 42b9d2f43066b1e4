#!/usr/bin/env python3
"""Kernel-trace companion of coresidency_probe.py: runs ONLY "scanline passes on one stream, right-view aggregation
on another" a few times (plus each alone once), for `rocprofv3 --kernel-trace`: the begin / end stamps of the
kernels say whether the scanline passes start while the aggregation runs and how long each takes beside it.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/coresidency_trace.py [reps]"""
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
Lf, Rf = Lu.float(), Ru.float()
adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0, placement_search=False, store_calibration=False)
dL, dR = torch.empty((H, W), device=DEV), torch.empty((H, W), device=DEV)
adc.ComputeBoth(dL, dR)
caL = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
caR = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
caL.ComputeArmLengths(Lu)
caR.ComputeArmLengths(Ru)
aggL, aggR, out = (torch.empty((H, W, D), device=DEV) for _ in range(3))
so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
caL.AggregationVertical(adc.GetPtrLeft(), aggL)
torch.cuda.synchronize()
# each alone
so.ScanLine(aggL, Lf, out, dL)
torch.cuda.synchronize()
caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    e = torch.cuda.Event()
    e.record()
    with torch.cuda.stream(s2):
        s2.wait_event(e)
        caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
    with torch.cuda.stream(s1):
        s1.wait_event(e)
        so.ScanLine(aggL, Lf, out, dL)
    torch.cuda.synchronize()
print("done")

#!/usr/bin/env python3
"""Times ScanlineOptimizer::ScanLine at 1920x1080 D=192 (HIP events, 3 rounds of 5) for the library named by
SMT_HIP_LIB -- A/B of scanline builds in separate processes on one box.  usage: python tools/scan_time.py [tag]"""
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
Lf = Lu.float()
adc = smt.AD_Census().Initialize(Lf, Ru.float(), D, H, W, 10.0, 30.0, placement_search=False, store_calibration=False)
adc.ComputeBoth()
ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
ca.ComputeArmLengths(Lu)
agg, out = torch.empty((H, W, D), device=DEV), torch.empty((H, W, D), device=DEV)
dL = torch.empty((H, W), device=DEV)
ca.AggregationVertical(adc.GetPtrLeft(), agg)
so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
res = {"tag": sys.argv[1] if len(sys.argv) > 1 else "", "lib": os.environ.get("SMT_HIP_LIB", "default"), "scanline_ms": []}
for _ in range(3):
    for _ in range(2):
        so.ScanLine(agg, Lf, out, dL)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        so.ScanLine(agg, Lf, out, dL)
    b.record()
    torch.cuda.synchronize()
    res["scanline_ms"].append(round(a.elapsed_time(b) / 5, 4))
print(json.dumps(res), flush=True)

#!/usr/bin/env python3
"""One config-3 front end (AD-Census, arms, left aggregation) and then `reps` ScanLine calls at 1920x1080 D=192:
the program rocprofv3 PMC passes are taken over (tools/prof_pmc_scanline.sh).  usage: scan_run.py [reps]"""
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
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    so.ScanLine(agg, Lf, out, dL)
torch.cuda.synchronize()
print("done")

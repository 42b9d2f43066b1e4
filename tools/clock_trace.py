#!/usr/bin/env python3
"""Time series of the headline kernel on one box: every ~second, 50 pairs through the ordinary path
(HIP-event kernel time), then smt_adcensus_diag (in-kernel shader clock, stamped kernel ms, store-only
ms on the same buffers).  Shows whether the cost kernel's 0.46 <-> 0.57 ms spread follows the clock, the
store ceiling, or neither.  Usage: python tools/clock_trace.py [seconds] [idle_gap_s]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereo_match_traditional_amd as smt  # noqa: E402
from stereo_match_traditional_amd import synth  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
dev = torch.device("cuda:0")
H, W, D = 1080, 1920, 192
L, R = synth.synth_pair(H, W, D, 3)
Lf = torch.from_numpy(L.astype(np.float32)).to(dev)
Rf = torch.from_numpy(R.astype(np.float32)).to(dev)
dl = torch.empty((H, W), device=dev)
dr = torch.empty((H, W), device=dev)
adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0)
t_start = time.perf_counter()
while time.perf_counter() - t_start < secs:
    adc.timing(1)
    t0 = time.perf_counter()
    for _ in range(50):
        adc.ComputeBoth(dl, dr)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e3
    prep, cost = adc.kernel_times()
    adc.timing(False)
    mhz, cms, sms = adc.diag(20)
    print(json.dumps({"t": round(time.perf_counter() - t_start, 2), "pair_ms": round(wall, 4),
                      "cost_ms": round(float(np.mean(cost)), 4), "cost_min": round(float(np.min(cost)), 4),
                      "tables_ms": round(float(np.mean(prep)), 4), "sclk_mhz": round(mhz, 1),
                      "stamped_cost_ms": round(cms, 4), "store_only_ms": round(sms, 4)}), flush=True)
    time.sleep(gap)

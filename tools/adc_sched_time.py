#!/usr/bin/env python3
"""A/B of the batch schedules of smt_adcensus_compute_batch on ONE handle in ONE process (the store speed of a handle's
volumes differs from process to process, DESIGN.md section 4, so schedules are compared on the same allocation):
SMT_OVERLAP = 0 in order, 1 tables on the internal stream, 2 table workgroups in the previous pair's cost launch.
Interleaved rounds, wall time per pair from HIP events around `steps` batch calls.
usage: python tools/adc_sched_time.py [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth

DEV = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = {}
for name, (H, W, D, B, steps) in {"1080p_d192": (1080, 1920, 192, 8, 10), "kitti_d256": (375, 1242, 256, 16, 10),
                                  "720p_d128": (720, 1280, 128, 8, 10)}.items():
    pairs = [synth.synth_pair(H, W, D, 3 + b) for b in range(B)]
    Lb = torch.stack([torch.from_numpy(p[0]).to(DEV).float() for p in pairs])
    Rb = torch.stack([torch.from_numpy(p[1]).to(DEV).float() for p in pairs])
    adc = smt.AD_Census().Initialize(Lb[0], Rb[0], D, H, W, 10.0, 30.0)
    dl = torch.empty((B, H, W), device=DEV); dr = torch.empty((B, H, W), device=DEV)
    out = {}
    for rnd in range(rounds):
        for sched in ("0", "1", "2"):
            os.environ["SMT_OVERLAP"] = sched
            for _ in range(2):
                adc.ComputeBatch(Lb, Rb, dl, dr)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(steps):
                adc.ComputeBatch(Lb, Rb, dl, dr)
            b.record()
            torch.cuda.synchronize()
            out.setdefault("sched%s_ms_per_pair" % sched, []).append(round(a.elapsed_time(b) / (steps * B), 4))
    adc.close()
    res[name] = out
print(json.dumps(res), flush=True)

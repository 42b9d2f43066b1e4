#!/usr/bin/env python3
"""Do the scanline passes run BESIDE the rectangle aggregation, or after it?  1920x1080 D=192 (configs[2]).
Times, with HIP events on the caller's stream: the left aggregation alone, the scanline alone, the right aggregation
alone, then scanline and right aggregation on two streams (the pipeline's overlap), and the whole batched entry --
for the aggregation occupancy given by SMT_AGG_WAVES (read by the library at handle creation).
usage: SMT_AGG_WAVES=4 python tools/coresidency_probe.py out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth

DEV = torch.device("cuda:0")
H, W, D = 1080, 1920, 192


def timed(fn, reps=5, warm=2):
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


def main():
    L, R = synth.synth_pair(H, W, D, 3)
    Lu, Ru = torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV)
    Lf, Rf = Lu.float(), Ru.float()
    adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0)
    dL, dR = torch.empty((H, W), device=DEV), torch.empty((H, W), device=DEV)
    adc.ComputeBoth(dL, dR)
    caL = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    caR = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    caL.ComputeArmLengths(Lu)
    caR.ComputeArmLengths(Ru)
    aggL, aggR, out = (torch.empty((H, W, D), device=DEV) for _ in range(3))
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
    res = {"SMT_AGG_WAVES": os.environ.get("SMT_AGG_WAVES"), "SMT_PIPE_SCHEDULE": os.environ.get("SMT_PIPE_SCHEDULE"),
           "lib": os.environ.get("SMT_HIP_LIB", "default")}
    res["aggregate_left_ms"] = timed(lambda: caL.AggregationVertical(adc.GetPtrLeft(), aggL))
    res["aggregate_right_ms"] = timed(lambda: caR.AggregationVertical(adc.GetPtrRight(), aggR, dR))
    res["scanline_ms"] = timed(lambda: so.ScanLine(aggL, Lf, out, dL))
    res["adcensus_ms"] = timed(lambda: adc.ComputeBoth(dL, dR))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        e = torch.cuda.Event()
        e.record()
        with torch.cuda.stream(s1):
            s1.wait_event(e)
            so.ScanLine(aggL, Lf, out, dL)
        with torch.cuda.stream(s2):
            s2.wait_event(e)
            caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)
    res["scanline_beside_aggregate_right_ms"] = timed(both)
    res["serial_sum_ms"] = res["scanline_ms"] + res["aggregate_right_ms"]

    def three():
        e = torch.cuda.Event()
        e.record()
        with torch.cuda.stream(s1):
            s1.wait_event(e)
            so.ScanLine(aggL, Lf, out, dL)
        with torch.cuda.stream(s2):
            s2.wait_event(e)
            caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
            caL.AggregationVertical(adc.GetPtrLeft(), aggR)        # stands for the next pair's left aggregation
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)
    res["scanline_beside_both_aggregations_ms"] = timed(three)
    for o in (adc, caL, caR, so):
        o.close()
    del aggL, aggR, out
    pipe = smt.Pipeline(H, W, D, DEV)
    L8, R8 = torch.stack([Lu] * 8), torch.stack([Ru] * 8)
    res["batched_entry_ms_per_pair"] = timed(lambda: pipe.run(L8, R8), reps=2, warm=1) / 8
    pipe.status()
    pipe.close()
    print(json.dumps(res))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()

"""Does the right-view aggregation overlap with the left view's aggregation + scanline when they
run on two HIP streams?  1920x1080 D=192.  usage: python tools/overlap_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth

DEV = "cuda:0"
H, W, D = 1080, 1920, 192
L, R = synth.synth_pair(H, W, D, 3)
Lu, Ru = torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV)
Lf = Lu.float()
adc = smt.AD_Census().Initialize(Lf, Ru.float(), D, H, W, 10, 30)
adc.ComputeBoth()
caL = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
caR = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
caL.ComputeArmLengths(Lu); caR.ComputeArmLengths(Ru)
aggL = torch.empty((H, W, D), device=DEV); aggR = torch.empty((H, W, D), device=DEV)
out = torch.empty((H, W, D), device=DEV)
dL = torch.empty((H, W), device=DEV); dR = torch.empty((H, W), device=DEV)
so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
s1h = torch.cuda.Stream(priority=-1)                    # high priority: the scanline's few waves should never wait for a slot

def serial():
    caL.AggregationVertical(adc.GetPtrLeft(), aggL)
    caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
    so.ScanLine(aggL, Lf, out, dL)

def two_streams():
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s1):
        s1.wait_event(e)
        caL.AggregationVertical(adc.GetPtrLeft(), aggL)
        so.ScanLine(aggL, Lf, out, dL)
    with torch.cuda.stream(s2):
        s2.wait_event(e)
        caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)

def right_first():
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s2):
        s2.wait_event(e)
        caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
    with torch.cuda.stream(s1):
        s1.wait_event(e)
        caL.AggregationVertical(adc.GetPtrLeft(), aggL)
        so.ScanLine(aggL, Lf, out, dL)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)

def staggered():
    """left aggregation alone, then the scanline (HBM / latency bound, 2 160 waves) beside the right aggregation
    (vector-issue bound)"""
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s1):
        s1.wait_event(e)
        caL.AggregationVertical(adc.GetPtrLeft(), aggL)
        e1 = torch.cuda.Event(); e1.record(s1)
        so.ScanLine(aggL, Lf, out, dL)
    with torch.cuda.stream(s2):
        s2.wait_event(e1)
        caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)


def staggered_prio():
    """as staggered, the scanline on a high-priority stream"""
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s1):
        s1.wait_event(e)
        caL.AggregationVertical(adc.GetPtrLeft(), aggL)
        e1 = torch.cuda.Event(); e1.record(s1)
    with torch.cuda.stream(s1h):
        s1h.wait_event(e1)
        so.ScanLine(aggL, Lf, out, dL)
    with torch.cuda.stream(s2):
        s2.wait_event(e1)
        caR.AggregationVertical(adc.GetPtrRight(), aggR, dR)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s1h); torch.cuda.current_stream().wait_stream(s2)


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t) / reps * 1e3, 3)

res = {}
for rnd in range(2):
    for name, fn in (("serial", serial), ("two_streams", two_streams), ("right_first", right_first), ("staggered", staggered), ("staggered_prio", staggered_prio)):
        res[f"{name}_{rnd}"] = timed(fn)
ref = (out.clone(), dL.clone(), dR.clone(), aggR.clone())
serial(); torch.cuda.synchronize()
res["equal"] = bool(torch.equal(ref[0], out) and torch.equal(ref[1], dL) and torch.equal(ref[2], dR) and torch.equal(ref[3], aggR))
print(json.dumps(res))

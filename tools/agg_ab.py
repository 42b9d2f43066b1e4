"""A/B of the aggregation kernel variants at 1920x1080 D=192 (left and right views), with a
bit-equality check against variant 0.  usage: python tools/agg_ab.py [--order 0|1]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--order", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--variants", default="3,4")
ap.add_argument("--sws", default="8,16,32,64")
ap.add_argument("--views", default="L,R")
ap.add_argument("--sweeps", default="0")
ap.add_argument("--size", default="1080,1920,192")
a = ap.parse_args()
DEV = "cuda:0"
H, W, D = [int(x) for x in a.size.split(",")]
L, R = synth.synth_pair(H, W, D, 3)
Lu, Ru = torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV)
adc = smt.AD_Census().Initialize(Lu.float(), Ru.float(), D, H, W, 10, 30)
adc.ComputeBoth()

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

res = {}
for name, img, vol in (("L", Lu, adc.GetPtrLeft()), ("R", Ru, adc.GetPtrRight())):
    if name not in a.views.split(","):
        continue
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(img)
    ref = torch.empty((H, W, D), device=DEV)
    out = torch.empty((H, W, D), device=DEV)
    fn = ca.AggregationVertical if a.order == 0 else ca.costAggregationV5
    ca.set_variant(0); ca.set_strip_width(16)
    res[f"{name}_v0_sw16"] = timed(lambda: fn(vol, ref), a.reps)
    for v in [int(x) for x in a.variants.split(",")]:
        ca.set_variant(v)
        for swp in [int(x) for x in a.sweeps.split(",")]:
            ca.set_sweep(swp)
            for sw in [int(x) for x in a.sws.split(",")]:
                ca.set_strip_width(sw)
                out.zero_()
                key = f"{name}_v{v}_sweep{swp}_sw{sw}"
                res[key] = timed(lambda: fn(vol, out), a.reps)
                if not torch.equal(out.view(torch.int32), ref.view(torch.int32)):
                    res[key + "_MISMATCH"] = True
    ca.close()
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))

"""CrossAggregator (a18) at 720p D=128, 4 iterations -- for rocprofv3 passes.  usage: crossagg_run.py [reps] [impl]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth
DEV = torch.device("cuda:0")
H, W, D = 720, 1280, 128
L, R = synth.synth_pair(H, W, D, 2)
bgr = torch.from_numpy(np.repeat(L[..., None], 3, axis=2).copy()).to(DEV)
cost = torch.rand((H, W, D), device=DEV)
ca = smt.CrossAggregator()
ca.Initialize(W, H, 0, D, DEV)
ca.SetData(bgr, bgr, cost)
ca.SetParams(34, 17, 20, 6)
if len(sys.argv) > 2:
    ca.set_impl(int(sys.argv[2]))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    ca.Aggregate(4)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); ca.Aggregate(4); b.record(); torch.cuda.synchronize()
arms = ca.arms().cpu().numpy().astype(np.int64) if hasattr(ca, "arms") else None
print("crossagg 4 iterations ms", a.elapsed_time(b), "mean arms (l r u d)", None if arms is None else arms.reshape(-1, 4).mean(0).tolist())

"""One ASW launch at config-4 size (960x540, D=128, 35x35 window, left view) -- for rocprofv3 passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth
DEV = torch.device("cuda:0")
H, W, D, ws = 540, 960, 128, 16
L, R = synth.synth_pair(H, W, D, 4)
Lp = torch.from_numpy(np.pad(L, ws + 1, mode="edge")).to(DEV)
Rp = torch.from_numpy(np.pad(R, ws + 1, mode="edge")).to(DEV)
sp, cm = smt.asw_masks(ws, 50.0, 30.0, DEV)
if len(sys.argv) > 2:
    smt.asw_set_impl(int(sys.argv[2]))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    d = smt.AdaptiveSupportWeight(Lp, Rp, ws, D, sp, cm, 40)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); d = smt.AdaptiveSupportWeight(Lp, Rp, ws, D, sp, cm, 40); b.record(); torch.cuda.synchronize()
print("asw ms", a.elapsed_time(b), "TFLOP/s f64", 8 * 35 * 35 * H * W * D / a.elapsed_time(b) / 1e9)

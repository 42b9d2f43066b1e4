"""NCC at NCC_main.cpp's window (21x21) on a 450x375 pair, D=64 -- timing / rocprofv3.  usage: ncc_run.py [reps] [impl] [D]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth
DEV = torch.device("cuda:0")
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
H, W, win = 375, 450, 10
L, R = synth.synth_pair(H, W, min(D, 64), 1)
Lt, Rt = torch.from_numpy(L).to(DEV), torch.from_numpy(R).to(DEV)
if len(sys.argv) > 2:
    smt.ncc_set_impl(int(sys.argv[2]))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    d = smt.NCC_algorithem(Lt, Rt, win, D)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); d = smt.NCC_algorithem(Lt, Rt, win, D); b.record(); torch.cuda.synchronize()
print("ncc ms", a.elapsed_time(b), "Mdisp/s", (H - 2 * win) * (W - 2 * win) * D / a.elapsed_time(b) / 1e3)

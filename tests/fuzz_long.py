"""Long randomized parity run (not collected by pytest): AD-Census both views + WTA and the
union-sharing aggregation kernels against the oracle at random shapes for ~150 s.
usage on the GPU box: python tests/fuzz_long.py [seed]   (last run: seed 777, 2541 + 2541 cases, all bit-exact)"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd._lib import QUIRK_FIX_RIGHT_ARM_STRIDE
from oracle import oracle as O
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12345
rng = np.random.default_rng(seed)
t0 = time.time(); n_adc = n_agg = 0
while time.time() - t0 < 150:
    # ---- AD-Census: both views + WTA at random shapes
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 300))
    D = int(rng.choice([1, 2, 7, 16, 63, 64, 65, 100, 128, 129, 191, 192, 193, 255, 256]))
    kind = rng.integers(0, 3)
    if kind == 0: L = rng.integers(0, 256, (H, W)); R = rng.integers(0, 256, (H, W))
    elif kind == 1: L = rng.integers(100, 104, (H, W)); R = rng.integers(100, 104, (H, W))
    else: L, R = O.synth_pair(H, W, max(D, 8), int(rng.integers(0, 1000)))
    L = L.astype(np.float32); R = R.astype(np.float32)
    adc = smt.AD_Census().Initialize(T(L), T(R), D, H, W, 10, 30)
    dl = torch.empty((H, W), device=DEV); dr = torch.empty((H, W), device=DEV)
    adc.ComputeBoth(dl, dr)
    vl = O.adcensus_view(L, R, D, 10.0, 30.0, 0); vr = O.adcensus_view(L, R, D, 10.0, 30.0, 1)
    assert np.array_equal(bits(adc.GetPtrLeft().cpu().numpy()), bits(vl)), ("adcL", H, W, D)
    assert np.array_equal(bits(adc.GetPtrRight().cpu().numpy()), bits(vr)), ("adcR", H, W, D)
    assert np.array_equal(dl.cpu().numpy(), O.wta(vl)) and np.array_equal(dr.cpu().numpy(), O.wta(vr)), ("wta", H, W, D)
    adc.close(); n_adc += 1
    # ---- aggregation, new variants
    H, W = int(rng.integers(2, 50)), int(rng.integers(2, 140))
    D = int(rng.choice([1, 5, 64, 100, 128, 192, 200, 256]))
    img = (rng.integers(0, 256, (H, W)) if rng.integers(0, 2) else (np.add.outer(np.arange(H) // 7, np.arange(W) // 19) * 23 % 200 + rng.integers(0, 4, (H, W)))).astype(np.uint8)
    order = int(rng.integers(0, 2)); chain = bool(rng.integers(0, 2)); tau = int(rng.choice([25, 30, 5]))
    arms = O.arms_all(img, tau, 6, 17, 34, chain=chain, right_row_bug=False)
    vol = rng.random((H, W, D), dtype=np.float32) * 2
    ref, oob = O.aggregate_rect(vol, arms, order)
    ca = smt.CrossArmAggregation().Initialize(H, W, tau, D, DEV, style="adcensus" if chain else "cblsm", quirks=QUIRK_FIX_RIGHT_ARM_STRIDE)
    ca.ComputeArmLengths(T(img))
    for variant in (3, 4, 5, 6):
        ca.set_variant(variant); ca.set_strip_width(int(rng.choice([8, 16, 32, 64])))
        out = torch.empty((H, W, D), device=DEV); disp = torch.empty((H, W), device=DEV)
        (ca.AggregationVertical if order == 0 else ca.costAggregationV5)(T(vol), out, disp)
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref)), ("agg", H, W, D, order, variant)
        assert np.array_equal(disp.cpu().numpy(), O.wta(ref)), ("aggwta", H, W, D, order, variant)
    ca.close(); n_agg += 1
print("fuzz ok: seed", seed, "adcensus cases", n_adc, "aggregation cases", n_agg)

"""Long randomized parity run (not collected by pytest): AD-Census both views + WTA (D up to 512), every union-sharing
aggregation kernel (variants 3-13, the matrix-pipe forms included), SAD (both formulations), ASW (all formulations), NCC (both
formulations), the scanline optimiser and CrossAggregator against the oracle at random shapes for ~150 s.
usage on the GPU box: python tests/fuzz_long.py [seed] [seconds]
(last runs, round 3: seed 4242, 240 s: 2 538 cases of each of the four families; seed 777, 300 s with the batch entry under
random schedules and aggregation variant 13: 1 202 cases of each; seed 2024, 400 s on the round's final code with NCC, the
scanline optimiser and CrossAggregator added: 1 634 cases of each family; all bit-exact)"""
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
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
n_sad = n_asw = n_more = 0
t_say = t0
while time.time() - t0 < budget:
    if time.time() - t_say > 60: t_say = time.time(); print("...", int(t_say - t0), "s", n_adc, "cases", flush=True)
    # ---- AD-Census: both views + WTA at random shapes
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 300))
    D = int(rng.choice([1, 2, 7, 16, 63, 64, 65, 100, 128, 129, 191, 192, 193, 255, 256, 257, 320, 400, 512]))
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
    # the same handle through the batch entry, 2..4 pairs (pair 0 = the pair above, the others its row-shifted / swapped
    # siblings), under a random schedule: in order, tables on the internal stream, tables inside the previous cost launch
    B = int(rng.integers(2, 5))
    Ls = [L] + [np.roll(L, b, axis=0) if b % 2 else R for b in range(1, B)]
    Rs = [R] + [np.roll(R, b, axis=0) if b % 2 else L for b in range(1, B)]
    os.environ["SMT_OVERLAP"] = str(int(rng.integers(0, 3)))
    dlb = torch.empty((B, H, W), device=DEV); drb = torch.empty((B, H, W), device=DEV)
    adc.ComputeBatch(T(np.stack(Ls)), T(np.stack(Rs)), dlb, drb)
    os.environ.pop("SMT_OVERLAP")
    for b in range(1, B):
        wl = O.adcensus_view(Ls[b], Rs[b], D, 10.0, 30.0, 0); wr = O.adcensus_view(Ls[b], Rs[b], D, 10.0, 30.0, 1)
        assert np.array_equal(dlb[b].cpu().numpy(), O.wta(wl)) and np.array_equal(drb[b].cpu().numpy(), O.wta(wr)), ("batch wta", H, W, D, b)
    assert np.array_equal(dlb[0].cpu().numpy(), O.wta(vl)) and np.array_equal(drb[0].cpu().numpy(), O.wta(vr)), ("batch wta 0", H, W, D)
    assert np.array_equal(bits(adc.GetPtrLeft().cpu().numpy()), bits(wl)) and np.array_equal(bits(adc.GetPtrRight().cpu().numpy()), bits(wr)), ("batch vol", H, W, D)
    adc.close(); n_adc += 1
    # ---- aggregation, new variants
    H, W = int(rng.integers(2, 50)), int(rng.integers(2, 140))
    D = int(rng.choice([1, 5, 64, 100, 128, 192, 200, 256, 300]))
    img = (rng.integers(0, 256, (H, W)) if rng.integers(0, 2) else (np.add.outer(np.arange(H) // 7, np.arange(W) // 19) * 23 % 200 + rng.integers(0, 4, (H, W)))).astype(np.uint8)
    order = int(rng.integers(0, 2)); chain = bool(rng.integers(0, 2)); tau = int(rng.choice([25, 30, 5]))
    arms = O.arms_all(img, tau, 6, 17, 34, chain=chain, right_row_bug=False)
    vol = rng.random((H, W, D), dtype=np.float32) * 2
    ref, oob = O.aggregate_rect(vol, arms, order)
    ca = smt.CrossArmAggregation().Initialize(H, W, tau, D, DEV, style="adcensus" if chain else "cblsm", quirks=QUIRK_FIX_RIGHT_ARM_STRIDE)
    ca.ComputeArmLengths(T(img))
    for variant in (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
        ca.set_variant(variant)
        if variant < 7: ca.set_strip_width(int(rng.choice([8, 16, 32, 64])))
        out = torch.empty((H, W, D), device=DEV); disp = torch.empty((H, W), device=DEV)
        (ca.AggregationVertical if order == 0 else ca.costAggregationV5)(T(vol), out, disp)
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref)), ("agg", H, W, D, order, variant)
        assert np.array_equal(disp.cpu().numpy(), O.wta(ref)), ("aggwta", H, W, D, order, variant)
    ca.close(); n_agg += 1
    # ---- SAD, both formulations, both views
    H, W = int(rng.integers(1, 24)), int(rng.integers(1, 150))
    D = int(rng.choice([1, 3, 20, 64, 65, 128, 200, 256])); ws = int(rng.integers(0, 5))
    L8 = rng.integers(0, 256, (H, W)).astype(np.uint8) if rng.integers(0, 2) else O.synth_pair(H, W, 16, int(rng.integers(0, 1000)))[0]
    R8 = rng.integers(0, 256, (H, W)).astype(np.uint8) if rng.integers(0, 2) else O.synth_pair(H, W, 16, int(rng.integers(0, 1000)))[1]
    Lp, Rp = O.pad_replicate(L8, ws + 1), O.pad_replicate(R8, ws + 1)
    rl, rr = O.sad(Lp, Rp, D, ws, 0), O.sad(Lp, Rp, D, ws, 1)
    for impl in (2, 1):
        smt.sad_set_impl(impl)
        assert np.array_equal(smt.GetPointDepthLeft(T(Lp), T(Rp), D, ws).cpu().numpy(), rl), ("sadL", H, W, D, ws, impl)
        assert np.array_equal(smt.GetPointDepthRight(T(Lp), T(Rp), D, ws).cpu().numpy(), rr), ("sadR", H, W, D, ws, impl)
    smt.sad_set_impl(2); n_sad += 1
    # ---- ASW, every formulation against the first (bit for bit) and the oracle's map, one view per case
    H, W = int(rng.integers(1, 14)), int(rng.integers(1, 90))
    D = int(rng.choice([1, 9, 64, 70, 130, 200, 256])); ws = int(rng.integers(1, 5)); view = int(rng.integers(0, 2))
    Lp, Rp = O.pad_replicate(L8[:H, :W] if L8.shape[0] >= H and L8.shape[1] >= W else rng.integers(0, 256, (H, W)).astype(np.uint8), ws + 1), None
    A8 = rng.integers(0, 256, (H, W)).astype(np.uint8); B8 = (A8.astype(np.int32) + rng.integers(-3, 4, (H, W))).clip(0, 255).astype(np.uint8)
    Lp, Rp = O.pad_replicate(A8, ws + 1), O.pad_replicate(B8, ws + 1)
    sp, cm = smt.asw_masks(ws, 50.0, 30.0, DEV); sp_ref, cm_ref = O.asw_masks(ws, 50.0, 30.0)
    rd = O.asw(Lp, Rp, D, ws, sp_ref, cm_ref, 40, view)
    ref_c = None
    for impl in (1, 3, 4, 5, 6):
        smt.asw_set_impl(impl)
        dd, cc = smt.AdaptiveSupportWeight(T(Lp), T(Rp), ws, D, sp, cm, 40, smt.VIEW_LEFT if view == 0 else smt.VIEW_RIGHT, want_cost=True)
        c = cc.cpu().numpy()
        if ref_c is None: ref_c = c
        ok = ~np.isnan(ref_c)
        assert np.array_equal(np.isnan(c), np.isnan(ref_c)) and np.array_equal(bits(c[ok]), bits(ref_c[ok])), ("asw", H, W, D, ws, view, impl)
        assert np.array_equal(dd.cpu().numpy(), rd), ("aswmap", H, W, D, ws, view, impl)
    smt.asw_set_impl(0); n_asw += 1
    # ---- NCC: both formulations against each other (costs to 1e-12, same NaN pattern) and the oracle's map
    H, W = int(rng.integers(3, 26)), int(rng.integers(3, 120))
    D = int(rng.choice([1, 5, 33, 61, 64, 65, 100, 125, 126, 128, 189, 190, 200, 253, 254, 256])); win = int(rng.integers(0, 6))
    Ln = rng.integers(0, 256, (H, W)).astype(np.uint8) if rng.integers(0, 2) else O.synth_pair(H, W, 16, int(rng.integers(0, 1000)))[0].copy()
    Rn = (Ln.astype(np.int32) + rng.integers(-2, 3, (H, W))).clip(0, 255).astype(np.uint8) if rng.integers(0, 2) else rng.integers(0, 256, (H, W)).astype(np.uint8)
    if H > 6 and W > 10: Ln[1:H // 2, 2:W // 2] = 77; Rn[0:H // 2 + 1, 0:W // 2 + 3] = 77        # flat patches: 0/0
    rd = O.ncc(Ln, Rn, D, win)
    got = []
    for impl in (2, 1):
        smt.ncc_set_impl(impl)
        dn, cn = smt.NCC_algorithem(T(Ln), T(Rn), win, D, want_cost=True)
        assert np.array_equal(dn.cpu().numpy(), rd), ("ncc map", H, W, D, win, impl)
        got.append(cn.cpu().numpy())
    smt.ncc_set_impl(2)
    if H > 2 * win and W > 2 * win:
        a, b = got[0][win:H - win, win:W - win], got[1][win:H - win, win:W - win]
        assert np.array_equal(np.isnan(a), np.isnan(b)), ("ncc nan", H, W, D, win)
        ok = ~np.isnan(a)
        assert not ok.any() or np.max(np.abs(a[ok] - b[ok])) <= 1e-12, ("ncc cost", H, W, D, win)
    # ---- scanline optimiser (4 passes + WTA) and CrossAggregator
    H, W = int(rng.integers(1, 14)), int(rng.integers(1, 60))
    D = int(rng.choice([1, 3, 16, 63, 64, 65, 128, 130, 192, 256, 300]))
    cost = rng.random((H, W, D), dtype=np.float32) * 3
    gray = rng.integers(0, 256, (H, W)).astype(np.float32)
    p1, p2 = int(rng.choice([10, 1, 0])), int(rng.choice([150, 3, 40]))
    so = smt.ScanlineOptimizer().Initialize(H, W, D, p1, p2, DEV)
    dsp = torch.empty((H, W), device=DEV)
    out = so.ScanLine(T(cost), T(gray), disp=dsp)
    ref = O.scanline(cost, gray, p1, p2)
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref)), ("scan", H, W, D, p1, p2)
    assert np.array_equal(dsp.cpu().numpy(), O.wta(ref)), ("scan wta", H, W, D)
    so.close()
    H, W = int(rng.integers(2, 30)), int(rng.integers(2, 50))
    D = int(rng.choice([1, 8, 64, 70, 192]))
    g8 = rng.integers(0, 256, (H, W)).astype(np.uint8) if rng.integers(0, 2) else O.synth_pair(H, W, 16, int(rng.integers(0, 1000)))[0]
    bgr = np.clip(g8[..., None].astype(np.int32) + rng.integers(0, 3, (H, W, 3)), 0, 255).astype(np.uint8)
    cst = (rng.random((H, W, D), dtype=np.float32) * np.exp2(rng.integers(-70, 70, D)).astype(np.float32)) if rng.integers(0, 2) else rng.random((H, W, D), dtype=np.float32)
    L1, L2, t1, t2, iters = int(rng.choice([34, 5, 60])), int(rng.choice([17, 2])), int(rng.choice([20, 8])), int(rng.choice([6, 3])), int(rng.integers(0, 5))
    a_ref, c_ref = O.crossagg(bgr, cst, L1, L2, t1, t2, iters)
    agg = smt.CrossAggregator(); assert agg.Initialize(W, H, 0, D, DEV)
    agg.SetData(T(bgr), T(bgr), T(cst)); agg.SetParams(L1, L2, t1, t2); agg.Aggregate(iters)
    assert np.array_equal(agg.get_arms_ptr().cpu().numpy(), a_ref), ("ca arms", H, W)
    assert np.array_equal(bits(agg.get_cost_ptr().cpu().numpy()), bits(c_ref)), ("ca cost", H, W, D, iters)
    agg.close(); n_more += 1
print("fuzz ok: seed", seed, "adcensus cases", n_adc, "aggregation cases", n_agg, "sad cases", n_sad, "asw cases", n_asw,
      "ncc + scanline + crossaggregator cases", n_more)

"""Seeded random-shape parity sweep: many small, odd shapes per stage, HIP path vs oracle, bit-exact.
Catches tile-boundary / lane-masking mistakes that the hand-picked cases might miss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rand_img(rng, H, W):
    kind = rng.integers(0, 3)
    if kind == 0:
        return rng.integers(0, 256, (H, W)).astype(np.uint8)
    if kind == 1:   # piecewise smooth
        base = (np.add.outer(np.arange(H) // int(rng.integers(3, 12)), np.arange(W) // int(rng.integers(5, 30))) * 17) % 200 + 20
        return (base + rng.integers(0, 3, (H, W))).astype(np.uint8)
    return np.full((H, W), int(rng.integers(0, 256)), np.uint8)


def test_fuzz_adcensus(smt, O):
    rng = np.random.default_rng(2024)
    for _ in range(40):
        H, W = int(rng.integers(1, 14)), int(rng.integers(1, 150))
        D = int(rng.choice([1, 2, 7, 16, 31, 60, 63, 64, 65, 100, 127, 128, 129, 191, 192, 200, 255, 256]))
        L, R = rand_img(rng, H, W), rand_img(rng, H, W)
        sc, ss = float(rng.choice([10.0, 3.5])), float(rng.choice([30.0, 11.0]))
        adc = smt.AD_Census().Initialize(T(L.astype(np.float32)), T(R.astype(np.float32)), D, H, W, sc, ss)
        dl = torch.empty((H, W), device=DEV)
        dr = torch.empty((H, W), device=DEV)
        adc.ComputeBoth(dl, dr)
        adc.status()
        ol, orr = O.adcensus_view(L, R, D, sc, ss, 0), O.adcensus_view(L, R, D, sc, ss, 1)
        tag = (H, W, D)
        assert np.array_equal(bits(adc.GetPtrLeft().cpu().numpy()), bits(ol)), tag
        assert np.array_equal(bits(adc.GetPtrRight().cpu().numpy()), bits(orr)), tag
        assert np.array_equal(dl.cpu().numpy(), O.wta(ol)), tag
        assert np.array_equal(dr.cpu().numpy(), O.wta(orr)), tag
        adc.close()


def test_fuzz_arms_and_aggregation(smt, O):
    rng = np.random.default_rng(7)
    for it in range(48):
        H, W = int(rng.integers(2, 60)), int(rng.integers(2, 120))
        D = int(rng.choice([1, 5, 16, 60, 64, 100, 128, 192]))
        img = rand_img(rng, H, W)
        order = int(rng.integers(0, 2))
        chain = bool(rng.integers(0, 2))
        tau = int(rng.choice([25, 30, 5]))
        # no stride bug here so that every shape is defined; the bug path has its own tests
        arms = O.arms_all(img, tau, 6, 17, 34, chain=chain, right_row_bug=False)
        vol = rng.random((H, W, D), dtype=np.float32) * 2
        ref, oob = O.aggregate_rect(vol, arms, order)
        assert oob == 0
        from stereo_match_traditional_amd._lib import QUIRK_FIX_RIGHT_ARM_STRIDE
        ca = smt.CrossArmAggregation().Initialize(H, W, tau, D, DEV, style="adcensus" if chain else "cblsm",
                                                  quirks=QUIRK_FIX_RIGHT_ARM_STRIDE)
        ca.set_variant(it % 14)
        ca.ComputeArmLengths(T(img))
        for g, r in zip(ca.arm_maps(), arms):
            assert np.array_equal(g.cpu().numpy(), r), (H, W, tau, chain)
        out = torch.empty((H, W, D), device=DEV)
        disp = torch.empty((H, W), device=DEV)
        (ca.AggregationVertical if order == 0 else ca.costAggregationV5)(T(vol), out, disp)
        ca.status()
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref)), (H, W, D, order, it % 6)
        assert np.array_equal(disp.cpu().numpy(), O.wta(ref))
        ca.close()


def test_fuzz_scanline_and_lrcheck(smt, O):
    rng = np.random.default_rng(99)
    for _ in range(30):
        H, W = int(rng.integers(1, 12)), int(rng.integers(1, 40))
        D = int(rng.choice([1, 3, 16, 63, 64, 65, 128, 130, 192, 256]))
        cost = rng.random((H, W, D), dtype=np.float32) * 3
        gray = rng.integers(0, 256, (H, W)).astype(np.float32)
        p1, p2 = int(rng.choice([10, 1, 0])), int(rng.choice([150, 3, 40]))
        so = smt.ScanlineOptimizer().Initialize(H, W, D, p1, p2, DEV)
        disp = torch.empty((H, W), device=DEV)
        out = so.ScanLine(T(cost), T(gray), disp=disp)
        ref = O.scanline(cost, gray, p1, p2)
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref)), (H, W, D, p1, p2)
        assert np.array_equal(disp.cpu().numpy(), O.wta(ref))
        so.close()
    for _ in range(30):
        H, W = int(rng.integers(1, 20)), int(rng.integers(1, 300))
        gate = int(rng.integers(0, 4))
        dL = rng.integers(0, 40, (H, W)).astype(np.float32)
        dR = rng.integers(0, 40, (H, W)).astype(np.float32)
        dL[rng.random((H, W)) < 0.05] = np.inf
        ref, cls, no, nm = O.lrcheck(dL, dR, gate)
        t = T(dL.copy())
        gcls, gno, gnm = smt.LeftRightConsistency(W, H, gate, t, T(dR))
        assert np.array_equal(gcls.cpu().numpy(), cls) and (gno, gnm) == (no, nm), (H, W, gate)
        assert np.array_equal(bits(t.cpu().numpy()), bits(ref))


def test_fuzz_sad_and_crossagg(smt, O):
    rng = np.random.default_rng(5)
    for _ in range(20):
        H, W = int(rng.integers(2, 16)), int(rng.integers(2, 50))
        D = int(rng.choice([1, 2, 16, 60, 64, 65, 130]))
        ws = int(rng.integers(0, 4))
        L, R = rand_img(rng, H, W), rand_img(rng, H, W)
        Lp, Rp = np.pad(L, ws + 1, mode="edge"), np.pad(R, ws + 1, mode="edge")
        assert np.array_equal(smt.GetPointDepthLeft(T(Lp), T(Rp), D, ws).cpu().numpy(), O.sad(Lp, Rp, D, ws, 0)), (H, W, D, ws)
        assert np.array_equal(smt.GetPointDepthRight(T(Lp), T(Rp), D, ws).cpu().numpy(), O.sad(Lp, Rp, D, ws, 1)), (H, W, D, ws)
    for _ in range(12):
        H, W = int(rng.integers(2, 40)), int(rng.integers(2, 60))
        D = int(rng.choice([1, 8, 64, 70, 192]))
        g = rand_img(rng, H, W)
        bgr = np.clip(g[..., None].astype(np.int32) + rng.integers(0, 3, (H, W, 3)), 0, 255).astype(np.uint8)
        cost = rng.random((H, W, D), dtype=np.float32)
        L1, L2 = int(rng.choice([34, 5, 60])), int(rng.choice([17, 2]))
        t1, t2 = int(rng.choice([20, 8])), int(rng.choice([6, 3]))
        iters = int(rng.integers(0, 5))
        a_ref, c_ref = O.crossagg(bgr, cost, L1, L2, t1, t2, iters)
        agg = smt.CrossAggregator()
        assert agg.Initialize(W, H, 0, D, DEV)
        agg.SetData(T(bgr), T(bgr), T(cost))
        agg.SetParams(L1, L2, t1, t2)
        agg.Aggregate(iters)
        assert np.array_equal(agg.get_arms_ptr().cpu().numpy(), a_ref), (H, W, L1, L2, t1, t2)
        assert np.array_equal(bits(agg.get_cost_ptr().cpu().numpy()), bits(c_ref)), (H, W, D, iters)
        agg.close()

"""CrossArm arms + aggregation, ScanlineOptimizer, LeftRightConsistency, CBLSM variants:
HIP path (through the C ABI) vs the CPU oracle, bit-exact.  Oracle = loop-for-loop
restatement of AD-CensusV1/{CrossArm.cpp,ScanlineOptimizer.h,PostProcessing.h} and
CBLSM/CBLSM.h ("parity unpinned", see oracle/smt_oracle.c)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def smooth_img(H, W, seed, step=3):
    """piecewise-smooth image with long flat runs so arms pass 17 and the sticky tau flips"""
    rng = np.random.default_rng(seed)
    base = (np.add.outer(np.arange(H) // 9, np.arange(W) // 23) * 17) % 200 + 20
    return (base + rng.integers(0, step, (H, W))).astype(np.uint8)


ARM_CASES = [
    # H, W, kind, seed
    (72, 160, "synth", 3),
    (72, 160, "noise", 4),      # never flips tau
    (64, 150, "smooth", 5),     # flips in the first direction
    (40, 200, "smooth", 6),
    (30, 64, "flat", 0),        # constant image: every arm saturates
]


def _img(H, W, kind, seed, O):
    if kind == "synth":
        return O.synth_pair(H, W, 32, seed)[0]
    if kind == "noise":
        return O.synth_pair(H, W, 32, seed, True)[0]
    if kind == "smooth":
        return smooth_img(H, W, seed)
    return np.full((H, W), 77, np.uint8)


@pytest.mark.parametrize("H,W,kind,seed", ARM_CASES)
@pytest.mark.parametrize("style", ["adcensus", "cblsm"])
def test_arms(smt, O, H, W, kind, seed, style):
    img = _img(H, W, kind, seed, O)
    ca = smt.CrossArmAggregation()
    if style == "adcensus":
        ref = O.arms_all(img, 30, 6, 17, 34, chain=True, right_row_bug=True)
        ca.Initialize(H, W, 30, 16, DEV)
    else:
        ref = O.arms_all(img, 25, 6, 17, 34, chain=False, right_row_bug=False)
        ca.Initialize(H, W, 25, 16, DEV, style="cblsm")
    ca.ComputeArmLengths(T(img))
    got = [a.cpu().numpy() for a in ca.arm_maps()]
    for name, g, r in zip("LRTB", got, ref):
        assert np.array_equal(g, r), f"arm {name} differs ({(g != r).sum()} px)"
    ca.close()


def test_arms_flip_inside_later_direction(smt, O):
    """tau must survive the left pass and flip in the top pass: columns are flat, rows are not."""
    H, W = 60, 140
    rng = np.random.default_rng(1)
    img = np.tile((rng.integers(0, 2, W) * 120 + 40).astype(np.uint8), (H, 1))
    ref = O.arms_all(img)
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, 8, DEV)
    ca.ComputeArmLengths(T(img))
    for g, r in zip(ca.arm_maps(), ref):
        assert np.array_equal(g.cpu().numpy(), r)
    ca.close()


def test_arms_three_channel(smt, O):
    H, W = 50, 120
    g = smooth_img(H, W, 9)
    img = O.synth_bgr(g, 4)
    ref = O.arms_all(img)
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, 8, DEV)
    ca.ComputeArmLengths(T(img))
    for a, r in zip(ca.arm_maps(), ref):
        assert np.array_equal(a.cpu().numpy(), r)
    ca.close()


AGG_CASES = [(72, 160, 16, "synth", 3), (64, 150, 64, "smooth", 5), (72, 160, 100, "noise", 4),
             (48, 180, 192, "synth", 8), (50, 183, 128, "smooth", 9), (40, 200, 256, "flat", 0)]


AGG_CASES += [(70, 155, 60, "smooth", 12), (66, 149, 7, "synth", 13)]   # W % 16 != 0, D % 4 != 0


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("H,W,D,kind,seed", AGG_CASES)
@pytest.mark.parametrize("order", [0, 1])
def test_aggregation(smt, O, H, W, D, kind, seed, order, variant):
    img = _img(H, W, kind, seed, O)
    vol = np.random.default_rng(seed).random((H, W, D), dtype=np.float32) * 2
    style = "adcensus" if order == 0 else "cblsm"
    if order == 0:
        arms = O.arms_all(img)
    else:
        arms = O.arms_all(img, 25, 6, 17, 34, chain=False, right_row_bug=False)
    ref, oob = O.aggregate_rect(vol, arms, order)
    assert oob == 0, "test size must not trigger the reference's out-of-plane reads"
    ca = smt.CrossArmAggregation().Initialize(H, W, 30 if order == 0 else 25, D, DEV, style=style)
    ca.set_variant(variant)
    ca.ComputeArmLengths(T(img))
    out = torch.empty((H, W, D), device=DEV)
    disp = torch.empty((H, W), device=DEV)
    (ca.AggregationVertical if order == 0 else ca.costAggregationV5)(T(vol), out, disp)
    ca.status()
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    assert np.array_equal(disp.cpu().numpy(), O.wta(ref))
    ca.close()


@pytest.mark.parametrize("order", [0, 1])
def test_aggregation_non_finite_inputs(smt, O, order):
    """inf / NaN in the input volume reach exactly the pixels whose rectangle holds them (the
    8-pixel kernel recomputes any pixel its flag arithmetic may have polluted)."""
    H, W, D = 60, 150, 64
    img = _img(H, W, "smooth", 6, O)
    rng = np.random.default_rng(6)
    vol = rng.random((H, W, D), dtype=np.float32) * 2
    for _ in range(12):
        i, j, d = int(rng.integers(0, H)), int(rng.integers(0, W)), int(rng.integers(0, D))
        vol[i, j, d] = [np.inf, -np.inf, np.nan][int(rng.integers(0, 3))]
    vol[30, 70, :] = np.inf
    vol[10, 20, 0] = np.nan
    vol[45, 100, 0] = -np.nan
    vol[20, 120, 37] = np.nan
    arms = O.arms_all(img) if order == 0 else O.arms_all(img, 25, 6, 17, 34, chain=False, right_row_bug=False)
    ref, oob = O.aggregate_rect(vol, arms, order)
    assert oob == 0
    outs = []
    for variant in (1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
        ca = smt.CrossArmAggregation().Initialize(H, W, 30 if order == 0 else 25, D, DEV,
                                                  style="adcensus" if order == 0 else "cblsm")
        ca.set_variant(variant)
        ca.ComputeArmLengths(T(img))
        out = torch.empty((H, W, D), device=DEV)
        disp = torch.empty((H, W), device=DEV)
        (ca.AggregationVertical if order == 0 else ca.costAggregationV5)(T(vol), out, disp)
        ca.status()
        outs.append((out.cpu().numpy(), disp.cpu().numpy()))
        ca.close()
    # the fused WTA follows `if (cost > value)` (CrossArm.cpp:44-52): a NaN never wins, a NaN at d = 0
    # freezes the result at 0
    dref = O.wta(ref)
    assert np.isnan(ref[..., 0]).any() and np.isnan(ref[..., 1:]).any()
    for o, dsp in outs:
        assert np.array_equal(np.isnan(o), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert np.array_equal(bits(o)[ok], bits(ref)[ok])
        assert np.array_equal(dsp, dref)


def test_aggregation_flags_reference_ub(smt, O):
    """Small image + stride bug -> rectangles leave the plane: reference UB, status says so."""
    from stereo_match_traditional_amd import SmtError
    H, W, D = 24, 40, 8
    img = np.full((H, W), 50, np.uint8)
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(T(img))
    out = torch.empty((H, W, D), device=DEV)
    ca.AggregationVertical(torch.rand((H, W, D), device=DEV), out)
    with pytest.raises(SmtError):
        ca.status()
    ca.close()


def test_cblsm_ad(smt, O):
    H, W, D = 20, 70, 60
    L, R = O.synth_pair(H, W, D, 5)
    for view, v in ((smt.VIEW_LEFT, 0), (smt.VIEW_RIGHT, 1)):
        got = smt.cblsm_ComputeAD(T(L), T(R), D, view).cpu().numpy()
        assert np.array_equal(bits(got), bits(O.cblsm_ad(L, R, D, v)))


SCAN_CASES = [(20, 40, 16, 3), (12, 70, 64, 4), (9, 33, 100, 5), (10, 50, 192, 6), (7, 21, 256, 7), (3, 5, 8, 8),
              (1, 9, 5, 9), (9, 1, 5, 10), (2, 2, 1, 11), (1, 1, 3, 12)]   # single row / column / hypothesis / pixel


@pytest.mark.parametrize("H,W,D,seed", SCAN_CASES)
def test_scanline_passes_and_sum(smt, O, H, W, D, seed):
    rng = np.random.default_rng(seed)
    cost = rng.random((H, W, D), dtype=np.float32) * 2
    gray = rng.integers(0, 256, (H, W)).astype(np.float32)
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
    for which in ("left", "right", "up", "down"):
        got = so.ScanPass(T(cost), T(gray), which).cpu().numpy()
        ref = O.scan_pass(cost, gray, 10, 150, which)
        assert np.array_equal(bits(got), bits(ref)), which
    disp = torch.empty((H, W), device=DEV)
    out = so.ScanLine(T(cost), T(gray), disp=disp).cpu().numpy()
    ref = O.scanline(cost, gray, 10, 150)
    assert np.array_equal(bits(out), bits(ref))
    assert np.array_equal(disp.cpu().numpy(), O.wta(ref))
    d2 = torch.empty((H, W), device=DEV)
    so.WTA(d2)
    assert np.array_equal(d2.cpu().numpy(), O.wta(ref))
    so.close()


def test_scanline_on_real_costs(smt, O):
    """ScanLine fed with AD-Census costs and image-derived p2 (smooth gray -> large p2)."""
    H, W, D = 16, 96, 64
    L, R = O.synth_pair(H, W, D, 11)
    cost = O.adcensus_view(L, R, D, 10.0, 30.0, 0)
    gray = L.astype(np.float32)
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
    out = so.ScanLine(T(cost), T(gray)).cpu().numpy()
    assert np.array_equal(bits(out), bits(O.scanline(cost, gray, 10, 150)))
    so.close()


@pytest.mark.parametrize("H,W,seed,gate", [(20, 60, 1, 2), (9, 200, 2, 1), (30, 31, 3, 5)])
def test_lrcheck(smt, O, H, W, seed, gate):
    rng = np.random.default_rng(seed)
    dL = rng.integers(0, 24, (H, W)).astype(np.float32)
    dR = rng.integers(0, 24, (H, W)).astype(np.float32)
    dL[rng.random((H, W)) < 0.05] = np.inf          # already-invalid inputs (:90-93)
    ref, cls, no, nm = O.lrcheck(dL, dR, gate)
    t = T(dL.copy())
    gcls, gno, gnm, occ, mis = smt.LeftRightConsistency(W, H, gate, t, T(dR), want_lists=True)
    assert np.array_equal(gcls.cpu().numpy(), cls)
    assert (gno, gnm) == (no, nm)
    assert np.array_equal(bits(t.cpu().numpy()), bits(ref))
    assert np.array_equal(occ, np.argwhere(cls == 1)) and np.array_equal(mis, np.argwhere(cls == 2))


def test_lrcheck_on_pipeline_maps(smt, O):
    H, W, D = 24, 120, 32
    L, R = O.synth_pair(H, W, D, 6)
    dl = O.wta(O.adcensus_view(L, R, D, 10.0, 30.0, 0))
    dr = O.wta(O.adcensus_view(L, R, D, 10.0, 30.0, 1))
    ref, cls, no, nm = O.lrcheck(dl, dr, 2)
    t = T(dl.copy())
    gcls, gno, gnm = smt.LeftRightConsistency(W, H, 2, t, T(dr))
    assert np.array_equal(gcls.cpu().numpy(), cls) and (gno, gnm) == (no, nm)
    assert np.array_equal(bits(t.cpu().numpy()), bits(ref))


def test_full_pipeline_config3_shape(smt, O):
    """main.cpp:57-92 order: AD-Census (L,R) -> arms+AggregationVertical (L on leftGray, R on
    rightGray) -> ScanLine on the left aggregated volume -> WTA -> LeftRightConsistency."""
    H, W, D = 72, 160, 64
    L, R = O.synth_pair(H, W, D, 3)
    # oracle
    cl = O.adcensus_view(L, R, D, 10.0, 30.0, 0)
    cr = O.adcensus_view(L, R, D, 10.0, 30.0, 1)
    al, oob1 = O.aggregate_rect(cl, O.arms_all(L), 0)
    ar, oob2 = O.aggregate_rect(cr, O.arms_all(R), 0)
    assert oob1 == 0 and oob2 == 0
    so_ref = O.scanline(al, L.astype(np.float32), 10, 150)
    dl_ref, dr_ref = O.wta(so_ref), O.wta(ar)
    lr_ref, cls_ref, no, nm = O.lrcheck(dl_ref, dr_ref, 2)
    # engine
    Lf, Rf = T(L.astype(np.float32)), T(R.astype(np.float32))
    adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0)
    adc.ComputeBoth()
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    aggL = torch.empty((H, W, D), device=DEV)
    aggR = torch.empty((H, W, D), device=DEV)
    dR = torch.empty((H, W), device=DEV)
    ca.ComputeArmLengths(T(L))
    ca.AggregationVertical(adc.GetPtrLeft(), aggL)
    ca.Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(T(R))
    ca.AggregationVertical(adc.GetPtrRight(), aggR, dR)
    ca.status()
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
    dL = torch.empty((H, W), device=DEV)
    out = so.ScanLine(aggL, Lf, disp=dL)
    assert np.array_equal(bits(out.cpu().numpy()), bits(so_ref))
    assert np.array_equal(dR.cpu().numpy(), dr_ref)
    cls, gno, gnm = smt.LeftRightConsistency(W, H, 2, dL, dR)
    assert np.array_equal(cls.cpu().numpy(), cls_ref) and (gno, gnm) == (no, nm)
    assert np.array_equal(bits(dL.cpu().numpy()), bits(lr_ref))


def test_config3_full_size_properties(smt):
    """configs[2] size (1920x1080, D=192): size-independent properties of the whole pipeline.
    - aggregation of a constant volume is that constant (mean of equal values, any rectangle);
    - fused WTAs equal the standalone WTA of the stored volumes;
    - scanline output >= 0 and its per-pixel minimum over d of each path is 0-shifted: out >= cost-ish;
    - LR check is idempotent (a second run rejects nothing new);
    - two runs give identical bits."""
    from stereo_match_traditional_amd import synth
    H, W, D = 1080, 1920, 192
    L, R = synth.synth_pair(H, W, D, 3)
    Lf, Rf = T(L.astype(np.float32)), T(R.astype(np.float32))
    adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0)
    adc.ComputeBoth()
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(T(L))
    agg = torch.empty((H, W, D), device=DEV)
    const = torch.full((H, W, D), 0.75, device=DEV)
    ca.AggregationVertical(const, agg)
    assert torch.equal(agg, const)
    dA = torch.empty((H, W), device=DEV)
    ca.AggregationVertical(adc.GetPtrLeft(), agg, dA)
    ca.status()
    assert torch.equal(smt.wta(agg), dA)
    assert float(agg.min()) >= 0.0 and float(agg.max()) < 2.0
    h1 = agg.view(torch.int32).sum(dtype=torch.int64).item()
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
    dS = torch.empty((H, W), device=DEV)
    out = so.ScanLine(agg, Lf, disp=dS)
    assert torch.equal(smt.wta(out), dS)
    assert float(out.min()) >= 0.0
    h2 = out.view(torch.int32).sum(dtype=torch.int64).item()
    # determinism
    ca.AggregationVertical(adc.GetPtrLeft(), agg, dA)
    assert agg.view(torch.int32).sum(dtype=torch.int64).item() == h1
    out2 = so.ScanLine(agg, Lf)
    assert out2.view(torch.int32).sum(dtype=torch.int64).item() == h2
    # LR check idempotence
    dR = torch.empty((H, W), device=DEV)
    smt.wta(adc.GetPtrRight(), dR)
    d1 = dS.clone()
    cls1, no1, nm1 = smt.LeftRightConsistency(W, H, 2, d1, dR)
    d2 = d1.clone()
    cls2, no2, nm2 = smt.LeftRightConsistency(W, H, 2, d2, dR)
    assert torch.equal(torch.isinf(d1), torch.isinf(d2))
    assert no1 + nm1 == int(torch.isinf(d1).sum())
    assert no2 + nm2 == no1 + nm1


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_cblsm_portrait_image(smt, O, variant):
    """CBLSM-style arms have no stride bug, so portrait images are defined: arms + row-major
    aggregation (costAggregationV5) on a 96x61 image, D not a multiple of 4."""
    H, W, D = 96, 61, 22
    img = smooth_img(H, W, 21)
    vol = np.random.default_rng(2).random((H, W, D), dtype=np.float32)
    arms = O.arms_all(img, 25, 6, 17, 34, chain=False, right_row_bug=False)
    ref, oob = O.aggregate_rect(vol, arms, 1)
    assert oob == 0
    ca = smt.CrossArmAggregation().Initialize(H, W, 25, D, DEV, style="cblsm")
    ca.set_variant(variant)
    ca.ComputeArmLengths(T(img))
    for g, r in zip(ca.arm_maps(), arms):
        assert np.array_equal(g.cpu().numpy(), r)
    out = torch.empty((H, W, D), device=DEV)
    disp = torch.empty((H, W), device=DEV)
    ca.costAggregationV5(T(vol), out, disp)
    ca.status()
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    assert np.array_equal(disp.cpu().numpy(), O.wta(ref))
    ca.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_aggregation_out_of_plane_taps_contribute_nothing(smt, O, variant):
    """Rectangles that leave the plane are reference UB (status says so); what every kernel
    variant then computes is documented: the sum over the in-plane taps, divided by the full
    rectangle area -- the oracle's own convention."""
    from stereo_match_traditional_amd import SmtError
    H, W, D = 24, 40, 64
    img = np.full((H, W), 50, np.uint8)
    vol = np.random.default_rng(variant).random((H, W, D), dtype=np.float32)
    arms = O.arms_all(img)
    ref, oob = O.aggregate_rect(vol, arms, 0)
    assert oob > 0
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.set_variant(variant)
    ca.ComputeArmLengths(T(img))
    out = torch.empty((H, W, D), device=DEV)
    disp = torch.empty((H, W), device=DEV)
    ca.AggregationVertical(T(vol), out, disp)
    with pytest.raises(SmtError):
        ca.status()
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    assert np.array_equal(disp.cpu().numpy(), O.wta(ref))
    ca.close()


def test_adcensus_style_portrait_is_rejected(smt):
    """With the reference's right-arm stride bug a portrait image makes it read outside the image."""
    from stereo_match_traditional_amd import SmtError
    ca = smt.CrossArmAggregation().Initialize(80, 50, 30, 8, DEV)
    with pytest.raises(SmtError):
        ca.ComputeArmLengths(torch.zeros((80, 50), dtype=torch.uint8, device=DEV))
    ca.close()


def test_config3_full_size_bands_vs_oracle(smt, O):
    """configs[2] size (1920x1080, D=192), stage by stage against the oracle on bands:
    - arms: the full maps (the oracle's sequential walk is cheap);
    - AggregationVertical: rows 500..519 of the d-slice 0..7, oracle fed with the band 460..559 of the
      GPU's own cost volume and the arm maps (rectangles of the compared rows stay inside the band);
    - ScanLineLeftRight: rows are independent, two rows of both passes."""
    from stereo_match_traditional_amd import synth
    H, W, D = 1080, 1920, 192
    L, R = synth.synth_pair(H, W, D, 3)
    Lf, Rf = T(L.astype(np.float32)), T(R.astype(np.float32))
    adc = smt.AD_Census().Initialize(Lf, Rf, D, H, W, 10.0, 30.0)
    adc.ComputeBoth()
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(T(L))
    arms_ref = O.arms_all(L)
    arms = [a.cpu().numpy() for a in ca.arm_maps()]
    for g, r in zip(arms, arms_ref):
        assert np.array_equal(g, r)
    agg = torch.empty((H, W, D), device=DEV)
    ca.AggregationVertical(adc.GetPtrLeft(), agg)
    ca.status()
    a, b = 460, 560
    sub = np.ascontiguousarray(adc.GetPtrLeft()[a:b, :, :8].cpu().numpy())
    ref, _ = O.aggregate_rect(sub, [m[a:b] for m in arms], 0)
    got = agg[500:520, :, :8].cpu().numpy()
    assert np.array_equal(bits(got), bits(ref[40:60]))
    so = smt.ScanlineOptimizer().Initialize(H, W, D, 10, 150, DEV)
    for which in ("left", "right"):
        out = so.ScanPass(agg, Lf, which)
        rows = slice(700, 702)
        ref = O.scan_pass(agg[rows].cpu().numpy(), L[rows].astype(np.float32), 10, 150, which)
        assert np.array_equal(bits(out[rows].cpu().numpy()), bits(ref)), which


@pytest.mark.parametrize("H,W,D,seed", [(40, 90, 16, 1), (33, 70, 64, 2), (25, 50, 100, 3)])
def test_cblsm_choose_arm_length(smt, O, H, W, D, seed):
    """CBLSM.h:65-236 (dead experiments): per-hypothesis arm volumes from the real arm
    maps of a synthetic pair, all four directions, against the oracle's loop restatement."""
    Li, Ri = O.synth_pair(H, W, 32, seed)
    aL = O.arms_all(Li, 25, 6, 17, 34, chain=False, right_row_bug=False)
    aR = O.arms_all(Ri, 25, 6, 17, 34, chain=False, right_row_bug=False)
    dev = [[T(a) for a in aL], [T(a) for a in aR]]
    (LL, LR, LU, LD), (RL, RR, RU, RD) = dev
    got = [smt.chooseArmLengthLeft(LL, LR, RL, RR, D, None, H, W),
           smt.chooseArmLengthRight(LL, LR, RL, RR, D, None, H, W),
           smt.chooseArmLengthUp(LU, LD, RU, RD, RL, RR, D, None, H, W),
           smt.chooseArmLengthDown(LU, LD, RU, RD, RL, RR, D, None, H, W)]
    ref = [O.choose_arm_length(0, aL[0], None, aR[0], aR[1], D),
           O.choose_arm_length(1, aL[1], None, aR[0], aR[1], D),
           O.choose_arm_length(2, aL[2], aR[2], aR[0], aR[1], D),
           O.choose_arm_length(3, aL[3], aR[3], aR[0], aR[1], D)]
    for name, g, r in zip(("Left", "Right", "Up", "Down"), got, ref):
        assert np.array_equal(g.cpu().numpy(), r), name


def test_aggregation_variants_agree_above_2gib(smt):
    """A 2.7 GB volume: tap byte offsets need all 32 bits (the buffer descriptor's num_records and the
    SGPR tap offsets are unsigned).  GPU-only property: the shared-tap kernels equal the plain walk."""
    H, W, D = 1300, 2000, 256
    g = torch.Generator(device=DEV).manual_seed(5)
    vol = torch.rand((H, W, D), device=DEV, generator=g) * 2
    img = (torch.arange(H, device=DEV)[:, None] // 11 * 9 + torch.arange(W, device=DEV)[None, :] // 17 * 13) % 200
    img = (img + torch.randint(0, 5, (H, W), device=DEV, generator=g)).to(torch.uint8)
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(img)
    ref = torch.empty((H, W, D), device=DEV)
    ca.set_variant(1)
    ca.AggregationVertical(vol, ref)
    ca.status()
    out = torch.empty((H, W, D), device=DEV)
    for variant in (13, 12, 11, 10, 9, 8, 7, 6, 4, 3, 0):
        out.zero_()
        ca.set_variant(variant)
        ca.AggregationVertical(vol, out)
        ca.status()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), variant
    ca.close()


def test_aggregation_mean_is_ieee_division_on_extreme_values(smt):
    """All aggregation variants must agree bit for bit on denormal, tiny, huge and ordinary sums alike
    (IEEE float division of the in-order sum by the rectangle area; a reciprocal-table shortcut that
    was tried failed exactly here, on sub-normal results)."""
    H, W, D = 80, 160, 64
    g = torch.Generator(device=DEV).manual_seed(11)
    # one scale per disparity plane (a rectangle sums within a plane), from denormal to near-overflow
    scale = torch.tensor([1e-44, 1e-41, 1e-39, 1e-38, 1e-30, 1e-3, 1.0, 3.0, 1e20, 1e34], device=DEV)
    vol = torch.rand((H, W, D), device=DEV, generator=g) * scale[torch.arange(D, device=DEV) % 10]
    img = ((torch.arange(H, device=DEV)[:, None] // 9 * 7 + torch.arange(W, device=DEV)[None, :] // 13 * 11) % 200
           + torch.randint(0, 6, (H, W), device=DEV, generator=g)).to(torch.uint8)
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.ComputeArmLengths(img)
    ref = torch.empty((H, W, D), device=DEV)
    ca.set_variant(1)
    ca.AggregationVertical(vol, ref)
    out = torch.empty((H, W, D), device=DEV)
    for variant in (13, 12, 11, 10, 9, 8, 7, 6, 4, 3, 5):
        out.zero_()
        ca.set_variant(variant)
        ca.AggregationVertical(vol, out)
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), variant
    assert torch.isfinite(ref).all() and ((ref > 0) & (ref < 1e-38)).any() and (ref > 1e30).any()
    ca.close()


@pytest.mark.parametrize("seed,max_arm", [(1, 6), (2, 20), (3, 34), (4, 60)])
def test_aggregation_fast_quotient_is_the_ieee_quotient(smt, seed, max_arm):
    """Variant 13 divides by the rectangle area with rcp + multiply + two FMAs where every value of a pixel lies in
    [2^-60, 2^61) and the area is at most 65535 (wave_quotient, smt_common.h); everywhere else, and in variant 1, the IEEE
    division runs.  Random rectangle areas from 1 to (2*max_arm+1)^2 (up to 12 480 on this image),
    random mantissas with exponents from -75 to 75 and both signs, a plane of zeros: bit-identical to variant 1."""
    H, W, D = 96, 130, 64
    g = torch.Generator(device=DEV).manual_seed(seed)
    ii = torch.arange(H, device=DEV)[:, None].expand(H, W); jj = torch.arange(W, device=DEV)[None, :].expand(H, W)
    def arms(limit):
        a = torch.randint(0, max_arm + 1, (H, W), device=DEV, generator=g)
        return torch.minimum(a, limit).to(torch.int32).contiguous()
    aL, aR, aT, aB = arms(jj), arms(W - 1 - jj), arms(ii), arms(H - 1 - ii)
    expo = torch.randint(-75, 76, (H, W, D), device=DEV, generator=g).float()
    sign = torch.randint(0, 2, (H, W, D), device=DEV, generator=g).float() * 2 - 1
    vol = (1 + torch.rand((H, W, D), device=DEV, generator=g)) * torch.exp2(expo) * sign
    # per-plane exponent for half of the planes (sums then stay near one magnitude), one plane of zeros
    plane = torch.exp2(torch.randint(-70, 70, (D,), device=DEV, generator=g).float())
    vol[:, :, ::2] = (1 + torch.rand((H, W, D // 2), device=DEV, generator=g)) * plane[::2]
    vol[:, :, 7] = 0.0
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
    ca.load_arm_maps(aL, aR, aT, aB)
    ref = torch.empty((H, W, D), device=DEV); out = torch.empty((H, W, D), device=DEV)
    for order, fn in ((0, ca.AggregationVertical), (1, ca.costAggregationV5)):
        ca.set_variant(1); fn(vol, ref)
        ca.set_variant(13); out.zero_(); fn(vol, out)
        ca.status()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), order
    area = ((aL + aR + 1) * (aT + aB + 1))
    assert int(area.min()) < 64 and (int(area.max()) > 8191) == (max_arm == 60)
    ca.close()


def test_wta_nan_semantics(smt, O):
    """smt_wta on arbitrary caller volumes: the reference's `if (cost > value)` (CrossArm.cpp:44-52,
    ScanlineOptimizer.h:51-59) is false for every comparison with a NaN, so a NaN never wins and a NaN at
    d = 0 freezes the result at 0; -0 ties +0; +-inf are ordinary values."""
    rng = np.random.default_rng(31)
    for D in (7, 64, 100, 192, 256):
        H, W = 16, 24
        vol = rng.standard_normal((H, W, D)).astype(np.float32)
        vol[0, 0, :] = np.nan                                  # all NaN -> 0
        vol[0, 1, 0] = np.nan                                  # NaN at d = 0 -> 0 whatever follows
        vol[0, 2, 0] = -np.nan
        vol[0, 3, :] = np.inf                                  # all +inf -> 0
        vol[0, 4, :] = np.inf; vol[0, 4, D - 1] = 5.0
        vol[0, 5, 1:] = np.nan                                 # only d = 0 is a number
        vol[0, 6, :] = 0.0; vol[0, 6, D // 2] = -0.0           # -0 does not beat +0
        vol[0, 7, :] = np.nan; vol[0, 7, 0] = 3.0; vol[0, 7, D - 1] = 2.0
        vol[0, 8, :] = -np.inf
        m = rng.random((H, W, D)) < 0.1
        m[0, :9] = False
        vol[m] = np.nan                                        # scattered NaNs of both signs
        m2 = rng.random((H, W, D)) < 0.05
        m2[0, :9] = False
        vol.view(np.uint32)[m2] = 0xFFC00001
        got = smt.wta(T(vol)).cpu().numpy()
        assert np.array_equal(got, O.wta(vol)), D


ARM_ORDERS = [[0, 1, 2, 3], [3, 2, 1, 0], [2, 0], [1], [0, 0, 3], [1, 3, 1, 0, 2]]


@pytest.mark.parametrize("kind,seed", [("synth", 3), ("smooth", 5), ("noise", 4)])
@pytest.mark.parametrize("order", ARM_ORDERS)
def test_arm_dir_calls_one_to_one(smt, O, kind, seed, order):
    """The four reference calls one by one (CrossArm.h:15-18) in any subset / order: the sticky member
    threshold is whatever the previous call left (CrossArm.cpp:223-225), maps not computed stay zero."""
    H, W = 64, 150
    img = _img(H, W, kind, seed, O)
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, 16, DEV)
    ca.Reset()
    ref = [np.zeros((H, W), np.int32) for _ in range(4)]
    tau = 30
    for d in order:
        _, tau = O.arms_dir(img, d, tau, out=ref[d])
        [ca.ComputeLeftArmLength, ca.ComputeRightArmLength, ca.ComputeTopArmLength, ca.ComputeButtonArmLength][d](T(img))
        assert ca.tao() == tau
    for g, r in zip(ca.arm_maps(), ref):
        assert np.array_equal(g.cpu().numpy(), r)
    if order == [0, 1, 2, 3]:
        for g, r in zip(ca.arm_maps(), O.arms_all(img)):
            assert np.array_equal(g.cpu().numpy(), r)
    # Reset = Initialize's state: threshold back, maps zeroed
    ca.Reset()
    assert ca.tao() == 30 and all(int(a.abs().sum()) == 0 for a in ca.arm_maps())
    ca.close()


def test_arm_dir_cblsm_threshold_is_by_value(smt, O):
    """CBLSM.h:643: `tao` is passed by value, so every call starts from tau again."""
    H, W = 64, 150
    img = smooth_img(H, W, 5)
    ca = smt.CrossArmAggregation().Initialize(H, W, 25, 16, DEV, style="cblsm")
    ca.Reset()
    for d in (3, 0):
        ca._arm_dir(T(img), d)
        assert ca.tao() == 25
    ref3, t = O.arms_dir(img, 3, 25, right_row_bug=False)
    ref0, _ = O.arms_dir(img, 0, 25, right_row_bug=False)
    assert t == 6          # the walk itself lowered its local copy
    maps = ca.arm_maps()
    assert np.array_equal(maps[3].cpu().numpy(), ref3) and np.array_equal(maps[0].cpu().numpy(), ref0)
    ca.close()


@pytest.mark.parametrize("H,W,D,kind", [(40, 90, 64, "smooth"), (2, 80, 20, "noise"), (30, 64, 192, "synth")])
def test_aggregation_exclusive_bounds_variant(smt, O, H, W, D, kind):
    """CrossArmAggregation::Aggregation (CrossArm.cpp:104-145, public but never called): rows outer,
    exclusive upper bounds; empty rectangles divide 0 by 0 -> NaN + SMT_ERR_REF_UB."""
    from stereo_match_traditional_amd import SmtError
    img = _img(H, W, kind, 9, O)
    vol = np.random.default_rng(H).random((H, W, D), dtype=np.float32)
    arms = O.arms_all(img, right_row_bug=False)
    ref, undefined = O.aggregate_rect(vol, arms, 2)
    ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV, quirks=1)
    ca.ComputeArmLengths(T(img))
    out = torch.empty((H, W, D), device=DEV)
    disp = torch.empty((H, W), device=DEV)
    ca.Aggregation(T(vol), out, disp)
    if undefined:
        with pytest.raises(SmtError):
            ca.status()
    else:
        ca.status()
    o = out.cpu().numpy()
    assert np.array_equal(np.isnan(o), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(bits(o)[ok], bits(ref)[ok])
    assert np.array_equal(disp.cpu().numpy(), O.wta(ref))
    ca.close()


@pytest.mark.parametrize("H,W,D,gate", [(40, 200, 64, 2.0), (33, 77, 16, 1.5), (5, 300, 100, 0.0)])
def test_left_and_right_consistency_variant(smt, O, H, W, D, gate):
    """LeftAndRightConsistency (PostProcessing.h:10-70): out of place, `>= gate` with a float gate, no
    +inf pre-check; non-finite disparities overflow the int conversion (x86: INT_MIN -> out of range)."""
    rng = np.random.default_rng(W)
    base = rng.integers(0, D, (H, 1)).astype(np.float32)
    dL = base + rng.integers(-3, 4, (H, W)).astype(np.float32)
    dR = base + rng.integers(-3, 4, (H, W)).astype(np.float32)
    dL[rng.random((H, W)) < 0.05] = np.inf
    dL[rng.random((H, W)) < 0.02] = np.nan
    dL[rng.random((H, W)) < 0.02] = -np.inf
    dL[rng.random((H, W)) < 0.02] = 3e9
    dR[rng.random((H, W)) < 0.03] = np.inf
    dR[rng.random((H, W)) < 0.02] = -4e9
    dR[rng.random((H, W)) < 0.02] = np.nan
    dL[rng.random((H, W)) < 0.1] += 0.5
    last_ref, cls_ref, no, nm = O.lrcheck_variant(dL, dR, gate)
    tl = T(dL)
    last = torch.full((H, W), -1.0, device=DEV)
    cls, go, gm = smt.LeftAndRightConsistency(tl, T(dR), last, W, H, gate)
    assert (go, gm) == (no, nm)
    assert np.array_equal(cls.cpu().numpy(), cls_ref)
    assert np.array_equal(bits(last.cpu().numpy()), bits(last_ref))
    assert np.array_equal(bits(tl.cpu().numpy()), bits(dL))          # leftDisp is only read


@pytest.mark.parametrize("H,W,D,win,seed", [(30, 70, 16, 1, 1), (24, 50, 40, 3, 2)])
def test_cblsm_cost_aggregation_new(smt, O, H, W, D, win, seed):
    """costAggregationNew + ComputeLocalValue (CBLSM.h:969-1045, :1087-1126; dead experiment)
    consuming the chooseArmLength* volumes, against the oracle's loop restatement, bit-exact."""
    Li, Ri = O.synth_pair(H, W, 32, seed)
    aL = O.arms_all(Li, 25, 6, 17, 34, chain=False, right_row_bug=False)
    aR = O.arms_all(Ri, 25, 6, 17, 34, chain=False, right_row_bug=False)
    vols = [O.choose_arm_length(0, aL[0], None, aR[0], aR[1], D), O.choose_arm_length(1, aL[1], None, aR[0], aR[1], D),
            O.choose_arm_length(2, aL[2], aR[2], aR[0], aR[1], D), O.choose_arm_length(3, aL[3], aR[3], aR[0], aR[1], D)]
    w = win + 1
    Lp, Rp = np.pad(Li, w, mode="edge"), np.pad(Ri, w, mode="edge")
    ref = O.cblsm_cost_aggregation_new(Lp, Rp, win, *vols)
    got = smt.costAggregationNew(T(Lp), T(Rp), None, *[T(v) for v in vols], D, H, W, win)
    assert np.array_equal(bits(got.cpu().numpy()), bits(ref))
    assert np.isfinite(ref).all() and ref.max() > 0


@pytest.mark.parametrize("kind,seed", [("synth", 3), ("smooth", 5), ("noise", 4), ("flat", 0)])
@pytest.mark.parametrize("sec,maxlen", [(17, 34), (5, 63), (0, 3), (20, 10), (40, 100)])
def test_arm_kernels_masks_vs_walk_vs_oracle(smt, O, kind, seed, sec, maxlen):
    """The bit-mask arm kernels (default for sec, maxlen <= 63), the neighbour-by-neighbour kernels
    (set_arm_walk; the only path beyond 63) and the oracle's sequential walk agree for ordinary and odd
    (sec = 0, maxlen < sec, long) parameters, gray and 3-channel, both threshold styles."""
    H, W = 70, 150
    img = _img(H, W, kind, seed, O)
    img3 = O.synth_bgr(img, seed + 1)
    for image in (img, img3):
        for style, tau, chain, bug in (("adcensus", 30, True, True), ("cblsm", 25, False, False)):
            ref = O.arms_all(image, tau, 6, sec, maxlen, chain=chain, right_row_bug=bug)
            for walk in (False, True):
                ca = smt.CrossArmAggregation().Initialize(H, W, tau, 16, DEV, style=style, sec_length=sec, max_length=maxlen)
                ca.set_arm_walk(walk)
                ca.ComputeArmLengths(T(image))
                for g, r in zip(ca.arm_maps(), ref):
                    assert np.array_equal(g.cpu().numpy(), r), (style, walk, image.ndim)
                ca.close()

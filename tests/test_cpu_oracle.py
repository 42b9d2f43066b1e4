"""CPU-only tests: the oracle against the reference build / golden fixtures, the host
logic, and the C-ABI export list.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_matches_oracle_generator(O):
    from stereo_match_traditional_amd import synth
    for (H, W, D, seed, noise) in [(24, 40, 16, 2, False), (33, 250, 64, 7, True), (50, 420, 192, 3, False)]:
        L0, R0 = O.synth_pair(H, W, D, seed, noise)
        L1, R1 = synth.synth_pair(H, W, D, seed, noise)
        assert np.array_equal(R0, R1) and np.array_equal(L0, L1)


def test_crossagg_oracle_vs_reference_build(O):
    """Pins orc_crossagg against the reference's own cross_aggregator.cpp (oracle/_ref)."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt .so)")
    for (H, W, D, seed, noise) in [(40, 56, 8, 7, False), (33, 47, 5, 11, True), (48, 64, 16, 3, False)]:
        L, _ = O.synth_pair(H, W, D, seed, noise)
        bgr = O.synth_bgr(L, seed + 5)
        cost = np.random.default_rng(seed).random((H, W, D), dtype=np.float32) * 2
        for iters in (1, 4):
            a0, c0 = O.crossagg(bgr, cost, iters=iters)
            a1, c1 = O.ref_crossagg(bgr, cost, iters=iters)
            assert np.array_equal(a0, a1)
            assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32))


def test_adcensus_closed_form_equals_faithful_loops(O):
    """Independent numpy restatement of SURVEY Appendix A.1-A.3 (census tables + LUT) vs
    the loop-for-loop oracle."""
    H, W, D = 13, 29, 20
    L, R = O.synth_pair(H, W, D, 5, True)
    Li, Ri = L.astype(np.int64), R.astype(np.int64)
    lutA, lutC = O.fuse_luts(10.0, 30.0)
    vol = np.zeros((H, W, D), np.float32)
    for i in range(H):
        for j in range(W):
            for d in range(D):
                x = j - d
                hd = 0
                for r in range(-4, 5):
                    for c in range(-3, 4):
                        if not (0 <= i + r < H and 0 <= j + c < W):
                            continue
                        lb = Li[i, j] > Li[i + r, j + c]
                        rb = Ri[i, max(x, 0)] > Ri[i + r, max(x + c, 0)]
                        hd += int(lb != rb)
                ad = abs(Li[i, j] - Ri[i, max(x, 0)])
                vol[i, j, d] = lutA[ad] + lutC[hd]
    ref = O.adcensus_view(L, R, D, 10.0, 30.0, 0)
    assert np.array_equal(vol.view(np.uint32), ref.view(np.uint32))


def test_lib_exports_every_declared_symbol():
    """libsmt_hip.so loads on CPU and exports everything include/smt.h declares."""
    from stereo_match_traditional_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "smt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = re.findall(r"\b(smt_[a-z0-9_]+)\s*\(", hdr)
    assert len(names) > 20
    missing = [n for n in sorted(set(names)) if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.smt_version() == 100


def test_fused_launch_grid_is_a_bijection():
    """smt_adcensus_compute_batch puts the table workgroups of pair n+1 into the grid of pair n's cost launch in groups
    of 8 (adcensus.hip: fused_grid / fused_decode).  The library's own host-side check of that arithmetic, over the
    benchmark shapes, tiny grids and random sizes: every cost group and every table group exactly once."""
    from stereo_match_traditional_amd import build
    lib = ctypes.CDLL(build.build())
    f = lib.smt_adcensus_selftest_fused_grid
    shapes = [(1080, 1920), (720, 1280), (375, 1242), (1, 1), (2, 70), (33, 64), (700, 3)]
    for H, W in shapes:
        nbx = (W + 63) // 64
        ncost = (nbx * H * 2 + 7) // 8 * 8
        ptx, pty, eb = (W + 63) // 64, (H + 31) // 32, (H + 15) // 16
        nprep = ptx * (pty + (eb + ptx - 1) // ptx)
        assert f(ncost, nprep) == 0, (H, W, ncost, nprep)
    rng = np.random.default_rng(3)
    for _ in range(3000):
        ncost, nprep = 8 * int(rng.integers(1, 4000)), int(rng.integers(1, 6000))
        assert f(ncost, nprep) == 0, (ncost, nprep)
    assert f(12, 5) != 0 and f(0, 5) != 0 and f(8, 0) != 0          # rejected arguments


def test_oracle_vs_golden_fixtures(O):
    """Oracle vs the committed fixtures (CrossAggregator outputs of the reference build;
    expf tables of the build container's libm)."""
    gold = os.path.join(ROOT, "tests", "golden")
    n = 0
    for f in sorted(os.listdir(gold)):
        if f.startswith("crossagg_") and f.endswith(".npz"):
            z = np.load(os.path.join(gold, f))
            L1, L2, t1, t2, iters = [int(v) for v in z["params"]]
            a, c = O.crossagg(z["bgr"], z["cost_init"], L1, L2, t1, t2, iters)
            assert np.array_equal(a, z["arms"])
            assert np.array_equal(c.view(np.uint32), z["cost_out"].view(np.uint32))
            n += 1
    assert n >= 3
    z = np.load(os.path.join(gold, "adcensus_luts_sc10_ss30.npz"))
    a, c = O.fuse_luts(10.0, 30.0)
    assert np.array_equal(a.view(np.uint32), z["lutA"].view(np.uint32))
    assert np.array_equal(c.view(np.uint32), z["lutC"].view(np.uint32))


def test_lrcheck_order_dependence_is_modelled(O):
    """A crafted row where the in-place write changes a later classification (:112 after :125)."""
    W = 12
    dL = np.zeros((1, W), np.float32)
    dR = np.zeros((1, W), np.float32)
    dL[0, 2] = 9       # cr = -7 -> out of range -> inf (mismatch)
    dL[0, 6] = 3       # cr = 3, dR[3] = -1... set so that crl = 2 (< 6) which is now inf
    dR[0, 3] = -1
    ref, cls, no, nm = O.lrcheck(dL, dR, 0)
    assert cls[0, 6] == 1 and np.isinf(ref[0, 2])    # reads inf at crl=2 -> occlusion


def test_c_abi_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching HIP (the reference has UB instead)."""
    import ctypes as C
    from stereo_match_traditional_amd._lib import lib, strerror
    L = lib()
    h = C.c_void_p()
    assert L.smt_adcensus_create(0, 10, 16, C.c_float(10), C.c_float(30), C.byref(h)) == -1
    assert L.smt_adcensus_create(10, 10, 513, C.c_float(10), C.c_float(30), C.byref(h)) == -1     # D > SMT_MAX_DISPARITY
    assert L.smt_adcensus_create(10, 10, 16, C.c_float(0), C.c_float(30), C.byref(h)) == -1
    assert L.smt_adcensus_compute(None, None, None, 3, None, None) == -1
    assert L.smt_wta(None, 4, 4, 4, None, None) == -1
    assert L.smt_crossarm_create(4, 4, 0, None, C.byref(h)) == -1
    assert L.smt_scanline_create(4, 4, 8, 10, 150, None) == -1
    assert L.smt_crossagg_create(0, 4, 8, C.byref(h)) == -1          # Initialize returns false (:28-31)
    assert L.smt_lrcheck(None, None, 4, 4, 2, None, None, None) == -1
    assert L.smt_sad(None, None, 4, 4, 8, 1, 1, None, None) == -1
    assert L.smt_ncc(None, None, 4, 4, 8, 1, None, None, None) == -1
    assert L.smt_asw(None, None, 4, 4, 8, 1, None, None, 40, 1, None, None, None) == -1
    assert strerror(-5).startswith("reference behaviour undefined")
    # host-only helpers work without a GPU
    sp = (C.c_double * 25)()
    cm = (C.c_double * 256)()
    assert L.smt_asw_masks(1, C.c_double(50), C.c_double(30), sp, cm) == 0
    assert abs(sp[12] - 1.0) < 1e-15 and cm[0] == 1.0
    cls = (C.c_uint8 * 6)(0, 1, 2, 2, 0, 1)
    occ = (C.c_int * 12)()
    mis = (C.c_int * 12)()
    no, nm = C.c_int(), C.c_int()
    assert L.smt_lrcheck_lists(cls, 2, 3, occ, C.byref(no), mis, C.byref(nm)) == 0
    assert (no.value, nm.value) == (2, 2) and list(occ[:4]) == [0, 1, 1, 2] and list(mis[:4]) == [0, 2, 1, 0]


def test_host_mirror_asw_masks_match_oracle(O):
    import ctypes as C
    from stereo_match_traditional_amd._lib import lib
    for ws, ss, sc in ((1, 50.0, 30.0), (11, 50.0, 30.0), (16, 10.0, 7.5)):
        side = 2 * ws + 3
        sp = np.empty((side, side), np.float64)
        cm = np.empty(256, np.float64)
        assert lib().smt_asw_masks(ws, C.c_double(ss), C.c_double(sc), sp.ctypes.data_as(C.c_void_p),
                                   cm.ctypes.data_as(C.c_void_p)) == 0
        rs, rc = O.asw_masks(ws, ss, sc)
        assert np.array_equal(sp, rs) and np.array_equal(cm, rc)


# ----------------------------------------------------------------------------------------------
# Independent restatements of SURVEY.md Appendix A (which the surveyor checked against a build of
# the reference) vs the loop-for-loop oracle.  Different formulation, same answers expected.
# ----------------------------------------------------------------------------------------------
def _arms_appendix_a4(img, tau0=30, low=6, sec=17, maxlen=34, row_bug=True):
    """A.4 in its PARALLEL form: per direction find the first row-major pixel whose neighbours
    1..17 are in-image and within tau0; pixels before it use tau0, it switches at k=18, later ones use 6;
    tau stays 6 for all later directions."""
    H, W = img.shape
    I = img.astype(np.int64)
    out = []
    tau_in = tau0
    for dirn in range(4):
        colR = H if (dirn == 1 and row_bug) else W
        def nb(i, j, k):
            if dirn == 0: return (i, j - k) if j - k >= 0 else None
            if dirn == 1: return (i, j + k) if j + k < colR else None
            if dirn == 2: return (i - k, j) if i - k >= 0 else None
            return (i + k, j) if i + k < H else None
        def far(i, j):
            return [j - 1 >= 1, j + 1 < colR - 1, i - 1 >= 1, i + 1 < H - 1][dirn]
        F = None
        if tau_in == tau0:
            for i in range(H):
                for j in range(colR):
                    ok = True
                    for k in range(1, sec + 1):
                        q = nb(i, j, k)
                        if q is None or abs(I[i, j] - I[q]) > tau0:
                            ok = False
                            break
                    if ok:
                        F = i * colR + j
                        break
                if F is not None:
                    break
        arm = np.zeros(H * W, np.int32)
        for i in range(H):
            for j in range(colR):
                idx = i * colR + j
                if tau_in == low or (F is not None and idx > F):
                    tA = tB = low
                elif F is not None and idx == F:
                    tA, tB = tau0, low
                else:
                    tA = tB = tau0
                saved, k = 0, 0
                while True:
                    k += 1
                    saved = k - 1
                    if k > sec and k > maxlen:
                        break
                    q = nb(i, j, k)
                    if q is None:
                        break
                    if abs(I[i, j] - I[q]) > (tB if k > sec else tA):
                        if far(i, j) and saved < 1:
                            saved = 1
                        break
                arm[idx] = saved
        out.append(arm.reshape(H, W))
        if F is not None:
            tau_in = low
    return out


def test_arms_parallel_form_equals_sequential_oracle(O):
    rng = np.random.default_rng(4)
    for kind in range(3):
        H, W = 26, 60
        if kind == 0:
            img = ((np.add.outer(np.arange(H) // 9, np.arange(W) // 23) * 17) % 200 + 20 + rng.integers(0, 3, (H, W))).astype(np.uint8)
        elif kind == 1:
            img = rng.integers(0, 256, (H, W)).astype(np.uint8)
        else:
            img = np.tile((rng.integers(0, 2, W) * 120 + 40).astype(np.uint8), (H, 1))   # flips only in the top pass
        ref = O.arms_all(img)
        got = _arms_appendix_a4(img)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)


def test_scanline_formula_a6_equals_oracle(O):
    """A.6 written as array code per line (not pointer walking)."""
    rng = np.random.default_rng(6)
    H, W, D, p1, p2i = 5, 7, 6, 10, 150
    C = (rng.random((H, W, D)) * 2).astype(np.float32)
    G = rng.integers(0, 256, (H, W)).astype(np.float32)
    f32 = np.float32

    def line(costs, grays, updown):
        n = len(costs)
        out = np.zeros((n, D), np.float32)
        out[0] = costs[0]
        last = np.full(D + 2, 65535, np.float32)
        last[1:D + 1] = costs[0]
        minlast = f32(min(last))
        gprev = grays[0]
        for s in range(1, n):
            p2 = max(f32(p1), f32(p2i) / f32(abs(f32(grays[s]) - f32(gprev)) + f32(1)))
            if not updown:
                gprev = grays[s]
            cur = np.zeros(D, np.float32)
            for d in range(D):
                l1 = last[d + 1]
                l2 = f32((last[d + 1] if updown else last[d]) + f32(p1))
                l3 = f32(last[d + 2] + f32(p1))
                l4 = f32(minlast + p2)
                cur[d] = f32(f32(costs[s][d] + min(min(l1, l2), min(l3, l4))) - minlast)
            out[s] = cur
            minlast = f32(min(f32(65535), cur.min()))
            last[1:D + 1] = cur
        return out

    Lv = np.zeros_like(C); Rv = np.zeros_like(C); Uv = np.zeros_like(C); Dv = np.zeros_like(C)
    flatG = G.reshape(-1)
    for i in range(H):
        Lv[i] = line(C[i], G[i], False)
        Rv[i] = line(C[i, ::-1], G[i, ::-1], False)[::-1]
    for j in range(W):
        Uv[:, j] = line(C[:, j], flatG[j:j + H], True)                        # gray pointer steps by ONE element
        start = (H - 1) * W + j
        Dv[:, j] = line(C[::-1, j], flatG[start - (H - 1):start + 1][::-1], True)[::-1]
    for name, v in (("left", Lv), ("right", Rv), ("up", Uv), ("down", Dv)):
        assert np.array_equal(v.view(np.uint32), O.scan_pass(C, G, p1, p2i, name).view(np.uint32)), name
    tot = ((Lv + Rv) + Uv) + Dv
    assert np.array_equal(tot.view(np.uint32), O.scanline(C, G, p1, p2i).view(np.uint32))


def test_lrcheck_two_phase_a7_equals_oracle(O):
    rng = np.random.default_rng(8)
    H, W, gate = 9, 40, 2
    dL = rng.integers(0, 20, (H, W)).astype(np.float32)
    dR = rng.integers(0, 20, (H, W)).astype(np.float32)
    dL[rng.random((H, W)) < 0.06] = np.inf
    ref, cls, no, nm = O.lrcheck(dL, dR, gate)

    def rejected(i, x):
        d = dL[i, x]
        if np.isinf(d):
            return True
        cr = int(np.float64(np.float32(x) - d) + 0.5)
        if 0 <= cr < W:
            return abs(d - dR[i, cr]) > gate
        return True
    got = np.zeros((H, W), np.uint8)
    for i in range(H):
        for j in range(W):
            d = dL[i, j]
            if np.isinf(d):
                got[i, j] = 2
                continue
            cr = int(np.float64(np.float32(j) - d) + 0.5)
            if not (0 <= cr < W):
                got[i, j] = 2
                continue
            dr = dR[i, cr]
            if abs(d - dr) > gate:
                crl = int(np.float64(np.float32(cr) + dr) + 0.5)
                if 0 < crl < W:
                    dl = np.inf if (crl < j and rejected(i, crl)) else dL[i, crl]
                    got[i, j] = 1 if dl > d else 2
                else:
                    got[i, j] = 2
    assert np.array_equal(got, cls)
    assert np.array_equal(np.isinf(ref), got != 0)


def test_aggregation_a5_equals_oracle(O):
    rng = np.random.default_rng(10)
    H, W, D = 72, 160, 3
    img = ((np.add.outer(np.arange(H) // 9, np.arange(W) // 23) * 17) % 200 + 20 + rng.integers(0, 3, (H, W))).astype(np.uint8)
    vol = rng.random((H, W, D), dtype=np.float32)
    arms = O.arms_all(img)
    ref, oob = O.aggregate_rect(vol, arms, 0)
    assert oob == 0
    flat = vol.reshape(H * W, D)
    f32 = np.float32
    for (i, j) in [(0, 0), (5, 17), (20, 100), (36, 159), (71, 3), (40, 80)]:
        Ll, Rr, up, dn = [int(a[i, j]) for a in arms]
        for d in range(D):
            v = f32(0)
            for l in range(-Ll, Rr + 1):
                for t in range(-up, dn + 1):
                    v = f32(v + flat[(i + t) * W + j + l, d])
            assert f32(v / f32((Ll + Rr + 1) * (up + dn + 1))) == ref[i, j, d]


def test_sad_a8_numpy_equals_oracle(O):
    """A.8: replicate-padded SAD, left-edge copy rule, OptimalDisparity; right view's unwritten last
    row/column and plain first-min."""
    H, W, D, winsize = 12, 30, 20, 1
    L, R = O.synth_pair(H, W, 32, 3)
    w = winsize + 1
    Lp, Rp = np.pad(L, w, mode="edge").astype(np.int64), np.pad(R, w, mode="edge").astype(np.int64)
    side = 2 * w + 1
    exp_l = np.zeros((H, W), np.int32)
    exp_r = np.zeros((H, W), np.int32)
    for i in range(H):
        for j in range(W):
            sad = np.zeros(D, np.float32)
            for d in range(D):
                dd = min(d, j)
                sad[d] = np.abs(Lp[i:i + side, j:j + side] - Rp[i:i + side, j - dd:j - dd + side]).sum()
            minv, best = np.float32(65535), np.float32(65535)
            for d in range(1, D):
                if minv > sad[d]:
                    minv, best = sad[d], np.float32(d)
            sec = sad[0]
            for d in range(D):
                if sad[d] != minv:
                    sec = min(sec, sad[d])
            if float(sec - minv) <= 0.01 or best == 0 or best == D - 1:
                exp_l[i, j] = 0
            else:
                exp_l[i, j] = int(best)
            if i < H - 1 and j < W - 1:
                sr = np.zeros(D, np.float32)
                for d in range(D):
                    dd = min(d, W - 1 - j)
                    sr[d] = np.abs(Lp[i:i + side, j + dd:j + dd + side] - Rp[i:i + side, j:j + side]).sum()
                exp_r[i, j] = int(np.argmin(sr))
    assert np.array_equal(O.sad(Lp.astype(np.uint8), Rp.astype(np.uint8), D, winsize, 0), exp_l)
    assert np.array_equal(O.sad(Lp.astype(np.uint8), Rp.astype(np.uint8), D, winsize, 1), exp_r)


def test_ncc_a9_numpy_equals_oracle(O):
    H, W, D, win = 14, 30, 12, 2
    L, R = O.synth_pair(H, W, 32, 4)
    L = L.copy(); R = R.copy()
    L[2:11, 4:20] = 77; R[2:11, 0:24] = 77            # flat -> NaN
    disp, cost = O.ncc(L, R, D, win, want_cost=True)
    Ld, Rd = L.astype(np.float64), R.astype(np.float64)
    for i in range(win, H - win):
        for j in range(win, W - win):
            c = np.empty(D)
            for d in range(D):
                if j - win - d >= 0:
                    a = Ld[i - win:i + win + 1, j - win:j + win + 1]
                    b = Rd[i - win:i + win + 1, j - win - d:j + win - d + 1]
                    x, y = a - a.sum() / a.size, b - b.sum() / b.size
                    with np.errstate(invalid="ignore", divide="ignore"):
                        c[d] = (x * y).sum() / (np.sqrt((x * x).sum()) * np.sqrt((y * y).sum()))
                else:
                    c[d] = 255.0
            ok = ~np.isnan(c)
            assert np.array_equal(np.isnan(cost[i, j]), ~ok)
            assert np.allclose(cost[i, j][ok], c[ok], rtol=0, atol=1e-9)
            best, m = 0, np.float32(cost[i, j][0])
            for d in range(1, D):
                if np.float64(m) < cost[i, j][d]:
                    best, m = d, np.float32(cost[i, j][d])
            assert disp[i, j] == best
    assert disp[:win].sum() == 0 and disp[:, :win].sum() == 0


def test_asw_a11_numpy_equals_oracle(O):
    H, W, D, ws, T = 8, 26, 10, 1, 40
    L, R = O.synth_pair(H, W, 32, 5)
    wins = ws + 1
    Lp, Rp = np.pad(L, wins, mode="edge"), np.pad(R, wins, mode="edge")
    sp, cm = O.asw_masks(ws, 50.0, 30.0)
    side = 2 * wins + 1
    c = (side - 1) // 2
    yy, xx = np.mgrid[0:side, 0:side]
    assert np.array_equal(sp, np.exp(-(((xx - c) ** 2 + (yy - c) ** 2).astype(np.float64)) / (2 * 50.0 * 50.0)))
    for view in (0, 1):
        disp, cost = O.asw(Lp, Rp, D, ws, sp, cm, T, view, want_cost=True)
        A, B = (Lp, Rp) if view == 0 else (Rp, Lp)
        Wp = W + 2 * wins
        for i in range(H):
            for j in range(W):
                a = A[i:i + side, j:j + side].astype(np.int64)
                dmax = j if view == 0 else W - wins - 2 - j
                if dmax < 0:
                    assert disp[i, j] == 0 and np.isnan(cost[i, j]).all()
                    continue
                cv = np.empty(D, np.float32)
                for d in range(D):
                    dd = min(d, dmax)
                    x0 = j - dd if view == 0 else j + dd
                    b = B[i:i + side, x0:x0 + side].astype(np.int64)
                    w0 = cm[np.abs(a - a[wins, wins])] * sp
                    w1 = cm[np.abs(b - b[wins, wins])] * sp
                    m2 = w0 * w1
                    e = np.minimum(np.abs(a - b), T)
                    cv[d] = np.float32((m2 * e).sum() / m2.sum())
                assert np.max(np.abs(cv - cost[i, j])) <= 1e-5
                # identical WTA unless two costs are within float noise of each other
                srt = np.sort(cv)
                if srt[1] - srt[0] > 1e-5:
                    assert disp[i, j] == int(np.argmin(cv))


def _py_fill_the_hole(disp, dispRange, occ, mis):
    """Independent restatement of FillTheHole (PostProcessing.h:156-248) in plain Python: lists
    walked in order, one mutable `angle`, writes after reads, the width/height swap."""
    import math
    d = np.array(disp, np.float32).reshape(-1).copy()
    row, col = disp.shape
    width, height = row, col
    f32 = np.float32
    pi = f32(3.1415926)
    a1 = [pi, f32(3) * pi / f32(4), pi / f32(2), pi / f32(4), f32(0), f32(7) * pi / f32(4), f32(3) * pi / f32(2), f32(5) * pi / f32(4)]
    a2 = [pi, f32(5) * pi / f32(4), f32(3) * pi / f32(2), f32(7) * pi / f32(4), f32(0), pi / f32(4), pi / f32(2), f32(3) * pi / f32(4)]
    angle = a1
    mis = [tuple(p) for p in mis]
    occ = [tuple(p) for p in occ]
    replaced = None

    def lround(v):                       # float argument, half away from zero
        v = float(v)
        return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))

    for k in range(3):
        trg = occ if k == 0 else mis
        if not trg:
            continue
        cap = len(trg)
        if k == 2:
            trg = [(i, j) for i in range(height) for j in range(width) if d[i * width + j] == 65535]
            mis = replaced = trg
            assert len(trg) <= cap, "reference UB"
        fill = [f32(0)] * len(trg)
        for n, (y, x) in enumerate(trg):
            if y == height // 2:
                angle = a2
            got = []
            for s in range(8):
                sina = np.sin(angle[s], dtype=np.float32)
                cosa = np.cos(angle[s], dtype=np.float32)
                for m in range(1, int(1.0 * dispRange)):
                    yy = lround(f32(y) + f32(m) * sina)
                    xx = lround(f32(x) + f32(m) * cosa)
                    if yy < 0 or yy >= height or xx < 0 or xx >= width:
                        break
                    v = d[yy * width + xx]
                    if v != 65535:
                        got.append(v)
                        break
            if not got:
                continue
            got.sort()
            fill[n] = (got[1] if len(got) > 1 else got[0]) if k == 0 else got[len(got) // 2]
        for n, (y, x) in enumerate(trg):
            d[y * width + x] = fill[n]
    return d.reshape(row, col), replaced


@pytest.mark.parametrize("row,col,seed", [(24, 24, 0), (20, 31, 1), (18, 40, 2)])
def test_fill_the_hole_python_restatement_equals_oracle(O, row, col, seed):
    """numpy's float32 sin/cos and glibc's sinf/cosf agree on these 16 angles (checked here too)."""
    rng = np.random.default_rng(seed)
    D = 16
    d = rng.integers(0, D, (row, col)).astype(np.float32)
    holes = rng.random((row, col)) < 0.15
    d[holes] = 65535
    d[rng.random((row, col)) < 0.03] = np.inf            # what LeftRightConsistency leaves behind
    n = row * col

    def pairs(k):
        flat = rng.integers(0, n, k)
        # (first, second) with first*row + second inside the buffer; first < col keeps it a line index
        return np.stack([flat // row, flat % row], 1).astype(np.int32)
    occ = pairs(30)
    mis = np.concatenate([pairs(int(holes.sum()) + 5), occ[:3]])      # includes duplicates
    mis[7, 0] = col // 2                                              # the angle switch, mid-list
    ref, third = O.fill_the_hole(d, D, occ, mis)
    got, third_py = _py_fill_the_hole(d, D, occ, mis)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    assert third is not None and [tuple(p) for p in third] == third_py
    # no mismatch list -> no third pass (:174): holes survive
    ref2, third2 = O.fill_the_hole(d, D, occ, np.empty((0, 2), np.int32))
    assert third2 is None and (ref2 == 65535).sum() > 0
    # more holes than mismatch entries: the reference overruns fill_disps
    with pytest.raises(ValueError):
        O.fill_the_hole(d, D, occ, mis[:3])


def test_choose_arm_length_closed_forms_equal_oracle(O):
    """Independent closed forms of CBLSM.h:65-147 (left / right) and a plain-Python walk of the
    up / down loops (:151-236) against the oracle's loop restatement."""
    rng = np.random.default_rng(3)
    row, col, D = 19, 27, 12
    jj = np.arange(col)[None, :].repeat(row, 0)
    ii = np.arange(row)[:, None].repeat(col, 1)
    LL = np.minimum(rng.integers(0, 9, (row, col)), jj).astype(np.int32)
    LR = np.minimum(rng.integers(0, 9, (row, col)), col - 1 - jj).astype(np.int32)
    RL = np.minimum(rng.integers(0, 9, (row, col)), jj).astype(np.int32)
    RR = np.minimum(rng.integers(0, 9, (row, col)), col - 1 - jj).astype(np.int32)
    LU = np.minimum(rng.integers(0, 9, (row, col)), ii).astype(np.int32)
    LD = np.minimum(rng.integers(0, 9, (row, col)), row - 1 - ii).astype(np.int32)
    RU = np.minimum(rng.integers(0, 9, (row, col)), ii).astype(np.int32)
    RD = np.minimum(rng.integers(0, 9, (row, col)), row - 1 - ii).astype(np.int32)
    d = np.arange(D)[None, None, :]
    # left: d must lie within [0, min(RL, RR)]; then the a with a + d <= RL, at most LL of them
    left = np.where((d > RL[..., None]) | (d > RR[..., None]), 0,
                    np.clip(np.minimum(LL[..., None], RL[..., None] - d), 0, None))
    assert np.array_equal(O.choose_arm_length(0, LL, None, RL, RR, D), left)
    # right: d <= RL; consecutive a = 1.. with a - d >= -RL (always, as d <= RL) and a - d < RR
    right = np.where(d > RL[..., None], 0, np.clip(np.minimum(LR[..., None], RR[..., None] + d - 1), 0, None))
    assert np.array_equal(O.choose_arm_length(1, LR, None, RL, RR, D), right)
    up = np.zeros((row, col, D), np.int32)
    dn = np.zeros((row, col, D), np.int32)
    for i in range(row):
        for j in range(col):
            for dd in range(D):
                s = 0
                for u in range(1, LU[i, j] + 1):
                    pr = i - u
                    if pr >= i - RU[i, j]:
                        if j - dd < 0:
                            break
                        if -RL[pr, j] < -dd < RR[pr, j]:
                            s += 1
                    else:
                        s = 0
                        break
                up[i, j, dd] = s
                s = 0
                for u in range(1, LD[i, j] + 1):
                    pr = i + u
                    if pr <= i + RD[i, j]:
                        if j - dd < 0:
                            s = 0
                            break
                        if -RL[pr, j] <= -dd <= RR[pr, j]:
                            s += 1
                    else:
                        break
                dn[i, j, dd] = s
    assert np.array_equal(O.choose_arm_length(2, LU, RU, RL, RR, D), up)
    assert np.array_equal(O.choose_arm_length(3, LD, RD, RL, RR, D), dn)


def test_oracle_is_clean_under_asan_ubsan():
    """the CPU restatement built with -fsanitize=address,undefined and run once over every
    stage on exact-size buffers (oracle/sanitize_main.c, `make -C oracle sanitize`)."""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "oracle sanitizer run clean" in r.stdout


def test_header_is_plain_c(tmp_path):
    """include/smt.h is the drop-in boundary: it must compile as C99 (no C++ or HIP types in the signatures) and a C
    program must link against libsmt_hip.so through it."""
    import subprocess
    from stereo_match_traditional_amd import build
    lib = build.build()
    src = tmp_path / "t.c"
    src.write_text('#include "smt.h"\n#include <stdio.h>\n'
                   'int main(void) { smt_crossarm_params p; smt_crossarm_default_params(&p);\n'
                   '  printf("%d %d %s\\n", smt_version(), p.tau, smt_strerror(SMT_ERR_REF_UB)); return SMT_MAX_DISPARITY == 512 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", str(src), "-I", os.path.join(ROOT, "include"),
                        "-L", os.path.dirname(lib), "-lsmt_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split()[:2] == ["100", "30"], (r.stdout, r.stderr)

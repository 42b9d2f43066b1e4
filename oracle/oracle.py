"""ctypes front-end of the CPU oracle (oracle/smt_oracle.c) and of the reference build in
oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (stereo_match_traditional_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsmt_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libcrossagg_ref.so")


def build(verbose=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


def _load():
    # SMT_ORACLE_OMP=1: the rows-/planes-/lines-parallel OpenMP build of the same file (identical
    # results, independent iterations only) -- used by the fixture generator and the all-core baseline
    path = os.path.join(_HERE, "libsmt_oracle_omp.so") if os.environ.get("SMT_ORACLE_OMP") == "1" else _LIB
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
        _lib.orc_fnv1a.restype = C.c_uint64
        _lib.orc_aggregate_rect.restype = C.c_long
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def fnv1a(a):
    a = np.ascontiguousarray(a)
    return int(lib().orc_fnv1a(_p(a), C.c_size_t(a.nbytes)))


# --------------------------------------------------------------------------- synth
def synth_pair(H, W, D, seed, noise=False):
    L = np.empty((H, W), np.uint8)
    R = np.empty((H, W), np.uint8)
    lib().orc_synth_pair(H, W, D, C.c_uint32(seed), int(noise), _p(L), _p(R))
    return L, R


def synth_bgr(gray, seed):
    """BGR for CrossAggregator = gray + per-channel (byte mod 3)."""
    H, W = gray.shape
    s = np.uint32(seed)
    # vectorised LCG is awkward; small images only -> python loop over a uint64 recurrence
    n = H * W * 3
    out = np.empty(n, np.int32)
    st = int(seed) & 0xFFFFFFFF
    for k in range(n):
        st = (st * 1664525 + 1013904223) & 0xFFFFFFFF
        out[k] = (st >> 24) % 3
    bgr = gray.astype(np.int32)[..., None] + out.reshape(H, W, 3)
    return np.clip(bgr, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------- AD-Census
def fuse_luts(sigmaC, sigmaS):
    a = np.empty(256, np.float32)
    c = np.empty(64, np.float32)
    lib().orc_fuse_luts(C.c_float(sigmaC), C.c_float(sigmaS), _p(a), _p(c))
    return a, c


def adcensus_view(L, R, D, sigmaC, sigmaS, view, i0=0, i1=None, out=None):
    """ComputeADcensus (view 0) / ComputeADcensusRight (view 1) over rows [i0,i1)."""
    L = _c(L, np.float32)
    R = _c(R, np.float32)
    H, W = L.shape
    if i1 is None:
        i1 = H
    if out is None:
        out = np.zeros((H, W, D), np.float32)
    rc = lib().orc_adcensus_view(_p(L), _p(R), H, W, D, C.c_float(sigmaC), C.c_float(sigmaS),
                                 view, i0, i1, _p(out))
    assert rc == 0
    return out


def wta(vol):
    vol = _c(vol, np.float32)
    H, W, D = vol.shape
    disp = np.empty((H, W), np.float32)
    lib().orc_wta(_p(vol), H, W, D, _p(disp))
    return disp


# --------------------------------------------------------------------------- arms + aggregation
def arms_all(img, tau0=30, tau_low=6, sec=17, maxlen=34, chain=True, right_row_bug=True):
    img = _c(img, np.uint8)
    if img.ndim == 2:
        H, W = img.shape
        ch = 1
    else:
        H, W, ch = img.shape
    arms = [np.empty((H, W), np.int32) for _ in range(4)]
    rc = lib().orc_arms_all(_p(img), H, W, ch, tau0, tau_low, sec, maxlen, int(chain),
                            int(right_row_bug), *[_p(a) for a in arms])
    if rc != 0:
        raise ValueError("reference behaviour undefined for this shape (H > W with the _row bug)")
    return arms


def aggregate_rect(vol, arms, order=0):
    vol = _c(vol, np.float32)
    H, W, D = vol.shape
    out = np.empty_like(vol)
    a = [_c(x, np.int32) for x in arms]
    oob = lib().orc_aggregate_rect(_p(vol), H, W, D, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]),
                                   order, _p(out))
    return out, int(oob)


# --------------------------------------------------------------------------- scanline
def scan_pass(cost, gray, p1, p2, which):
    """which: 'left','right','up','down' -> one path volume."""
    cost = _c(cost, np.float32)
    gray = _c(gray, np.float32)
    H, W, D = cost.shape
    out = np.zeros_like(cost)
    if which in ("left", "right"):
        lib().orc_scan_lr(_p(cost), _p(gray), H, W, D, p1, p2, int(which == "left"), _p(out))
    else:
        lib().orc_scan_ud(_p(cost), _p(gray), H, W, D, p1, p2, int(which == "up"), _p(out))
    return out


def scanline(cost, gray, p1=10, p2=150):
    cost = _c(cost, np.float32)
    gray = _c(gray, np.float32)
    H, W, D = cost.shape
    out = np.empty_like(cost)
    rc = lib().orc_scanline(_p(cost), _p(gray), H, W, D, p1, p2, _p(out))
    assert rc == 0
    return out


def lrcheck(dL, dR, gate=2):
    dL = np.array(dL, np.float32, copy=True, order="C")
    dR = _c(dR, np.float32)
    H, W = dL.shape
    cls = np.empty((H, W), np.uint8)
    no = C.c_long()
    nm = C.c_long()
    lib().orc_lrcheck(_p(dL), _p(dR), H, W, gate, _p(cls), C.byref(no), C.byref(nm))
    return dL, cls, no.value, nm.value


def lrcheck_variant(dL, dR, gate):
    """LeftAndRightConsistency (PostProcessing.h:10-70) -> (lastDisp, cls, n_occ, n_mis)."""
    dL = _c(dL, np.float32)
    dR = _c(dR, np.float32)
    H, W = dL.shape
    last = np.empty((H, W), np.float32)
    cls = np.empty((H, W), np.uint8)
    no = C.c_long()
    nm = C.c_long()
    lib().orc_lrcheck_variant(_p(dL), _p(dR), _p(last), H, W, C.c_float(gate), _p(cls), C.byref(no), C.byref(nm))
    return last, cls, no.value, nm.value


def arms_dir(img, dirn, tau, tau_low=6, sec=17, maxlen=34, right_row_bug=True, out=None):
    """One Compute*ArmLength call (dirn 0 left, 1 right, 2 top, 3 bottom) entered with threshold `tau`;
    returns (map, threshold afterwards).  `out`: the map to write into (the right-arm call with the
    stride bug leaves most of it untouched); default = a zeroed map, as after Initialize."""
    img = _c(img, np.uint8)
    if img.ndim == 2:
        H, W = img.shape
        ch = 1
    else:
        H, W, ch = img.shape
    if out is None:
        out = np.zeros((H, W), np.int32)
    t = C.c_int(tau)
    lib().orc_arms_dir(_p(img), H, W, ch, int(dirn), C.byref(t), tau_low, sec, maxlen,
                       int(right_row_bug and dirn == 1), _p(out))
    return out, t.value


# --------------------------------------------------------------------------- CrossAggregator
def crossagg(bgr, cost_init, L1=34, L2=17, t1=20, t2=6, iters=4):
    bgr = _c(bgr, np.uint8)
    cost_init = _c(cost_init, np.float32)
    H, W, D = cost_init.shape
    arms = np.empty((H, W, 4), np.uint8)
    out = np.empty_like(cost_init)
    rc = lib().orc_crossagg(_p(bgr), _p(cost_init), W, H, D, L1, L2, t1, t2, iters, _p(arms), _p(out))
    assert rc == 0
    return arms, out


def have_ref():
    return os.path.exists(_REF)


def ref_crossagg(bgr, cost_init, L1=34, L2=17, t1=20, t2=6, iters=4):
    """The REFERENCE's own CrossAggregator (oracle/_ref, built from /root/reference)."""
    r = C.CDLL(_REF)
    bgr = _c(bgr, np.uint8)
    cost_init = _c(cost_init, np.float32)
    H, W, D = cost_init.shape
    arms = np.empty((H, W, 4), np.uint8)
    out = np.empty_like(cost_init)
    rc = r.ref_crossagg(_p(bgr), _p(cost_init), W, H, D, L1, L2, t1, t2, iters, _p(arms), _p(out))
    assert rc == 0
    return arms, out


# --------------------------------------------------------------------------- CBLSM
def cblsm_ad(L, R, D, view):
    L = _c(L, np.uint8)
    R = _c(R, np.uint8)
    H, W = L.shape
    out = np.zeros((H, W, D), np.float32)
    lib().orc_cblsm_ad(_p(L), _p(R), H, W, D, view, _p(out))
    return out


# --------------------------------------------------------------------------- window matchers
def pad_replicate(img, pad):
    return np.pad(img, pad, mode="edge")


def sad(Lp, Rp, D, winsize, view):
    Lp = _c(Lp, np.uint8)
    Rp = _c(Rp, np.uint8)
    Hp, Wp = Lp.shape
    w = winsize + 1
    disp = np.zeros((Hp - 2 * w, Wp - 2 * w), np.int32)
    lib().orc_sad(_p(Lp), _p(Rp), Hp, Wp, D, winsize, view, _p(disp))
    return disp


def sad_crosscheck(dL, dR):
    dL = _c(dL, np.int32)
    dR = _c(dR, np.int32)
    H, W = dL.shape
    out = np.empty((H, W), np.int32)
    cls = np.empty((H, W), np.uint8)
    lib().orc_sad_crosscheck(_p(dL), _p(dR), H, W, _p(out), _p(cls))
    return out, cls


def ncc(L, R, D, win, i0=0, i1=None, want_cost=False):
    L = _c(L, np.uint8)
    R = _c(R, np.uint8)
    H, W = L.shape
    if i1 is None:
        i1 = H
    disp = np.zeros((H, W), np.int32)
    cost = np.full((H, W, D), np.nan, np.float64) if want_cost else None
    lib().orc_ncc(_p(L), _p(R), H, W, D, win, i0, i1, _p(disp), _p(cost) if want_cost else None)
    return (disp, cost) if want_cost else disp


def asw_masks(winSize, sigma_s, sigma_c):
    side = 2 * winSize + 3
    sp = np.empty((side, side), np.float64)
    cm = np.empty(256, np.float64)
    lib().orc_asw_masks(winSize, C.c_double(sigma_s), C.c_double(sigma_c), _p(sp), _p(cm))
    return sp, cm


def asw(Lp, Rp, D, winSize, space, color, T, view, i0=0, i1=None, want_cost=False):
    Lp = _c(Lp, np.uint8)
    Rp = _c(Rp, np.uint8)
    Hp, Wp = Lp.shape
    wins = winSize + 1
    H, W = Hp - 2 * wins, Wp - 2 * wins
    if i1 is None:
        i1 = H
    disp = np.zeros((H, W), np.float32)
    cost = np.full((H, W, D), np.nan, np.float32) if want_cost else None
    space = _c(space, np.float64)
    color = _c(color, np.float64)
    lib().orc_asw(_p(Lp), _p(Rp), Hp, Wp, D, winSize, _p(space), _p(color), T, view, i0, i1,
                  _p(disp), _p(cost) if want_cost else None)
    return (disp, cost) if want_cost else disp


def asw_crosscheck(dL, dR):
    dL = _c(dL, np.float32)
    dR = _c(dR, np.float32)
    H, W = dL.shape
    out = np.empty((H, W), np.uint8)
    lib().orc_asw_crosscheck(_p(dL), _p(dR), H, W, _p(out))
    return out


# --------------------------------------------------------------------------- staging / post
def bgr2gray(bgr):
    bgr = _c(bgr, np.uint8)
    H, W, _ = bgr.shape
    g = np.empty((H, W), np.uint8)
    lib().orc_bgr2gray(_p(bgr), H * W, _p(g))
    return g


def median(inp, wnd):
    inp = _c(inp, np.float32)
    H, W = inp.shape
    out = np.empty_like(inp)
    lib().orc_median(_p(inp), _p(out), W, H, wnd)
    return out


def remove_speckles(d, diff, min_area, invalid_val):
    d = np.array(d, np.float32, copy=True, order="C")
    H, W = d.shape
    lib().orc_remove_speckles(_p(d), W, H, int(diff), C.c_uint(min_area), int(invalid_val))
    return d


def fill_the_hole(disp, dispRange, occ, mis):
    """FillTheHole on a [row][col] map.  Returns (filled map, third-pass list or None); raises
    ValueError where the reference writes out of bounds."""
    d = np.array(disp, np.float32, copy=True, order="C")
    row, col = d.shape
    occ = np.ascontiguousarray(np.asarray(occ, np.int32).reshape(-1, 2))
    mis = np.ascontiguousarray(np.asarray(mis, np.int32).reshape(-1, 2))
    third = np.empty((row * col, 2), np.int32)
    nt = C.c_int(-1)
    rc = lib().orc_fill_the_hole(_p(d), row, col, int(dispRange), _p(occ), len(occ), _p(mis), len(mis),
                                 _p(third), C.byref(nt))
    if rc != 0:
        raise ValueError("reference behaviour undefined (out-of-bounds write in FillTheHole)")
    return d, (third[:nt.value].copy() if nt.value >= 0 else None)


def choose_arm_length(dirn, own, vert, RL, RR, D):
    """chooseArmLength{Left,Right,Up,Down} (dirn 0..3) -> int32 [row][col][D]."""
    own = _c(own, np.int32); RL = _c(RL, np.int32); RR = _c(RR, np.int32)
    vert = _c(vert, np.int32) if vert is not None else None
    row, col = own.shape
    vol = np.empty((row, col, D), np.int32)
    lib().orc_choose_arm_length(int(dirn), _p(own), _p(vert) if vert is not None else None, _p(RL), _p(RR),
                                row, col, int(D), _p(vol))
    return vol


def cblsm_cost_aggregation_new(Lp, Rp, winSize, armL, armR, armUp, armDown):
    """costAggregationNew (CBLSM.h:1087-1126) on padded images + int32 [H][W][D] arm volumes."""
    Lp = _c(Lp, np.uint8); Rp = _c(Rp, np.uint8)
    Hp, Wp = Lp.shape
    vols = [_c(a, np.int32) for a in (armL, armR, armUp, armDown)]
    H, W, D = vols[0].shape
    assert H == Hp - 2 * (winSize + 1) and W == Wp - 2 * (winSize + 1)
    cost = np.empty((H, W, D), np.float32)
    lib().orc_cblsm_cost_aggregation_new(_p(Lp), _p(Rp), Hp, Wp, int(winSize), *[_p(v) for v in vols], int(D), _p(cost))
    return cost

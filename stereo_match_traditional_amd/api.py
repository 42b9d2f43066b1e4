"""Host-side mirrors of the reference's classes / free functions over the C ABI.

Names and argument meaning follow the reference (file:line cited per class); buffers are
torch CUDA(=HIP) tensors whose data_ptr() is handed to libsmt_hip.so.  Nothing here
computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, VIEW_LEFT, VIEW_RIGHT, VIEW_BOTH

__all__ = ["FillTheHole", "chooseArmLengthLeft", "chooseArmLengthRight", "chooseArmLengthUp", "chooseArmLengthDown", "costAggregationNew", "AD_Census", "wta", "current_stream_ptr", "CrossArmAggregation", "cblsm_ComputeAD",
           "ScanlineOptimizer", "LeftRightConsistency", "LeftAndRightConsistency", "CrossAggregator", "GetPointDepthLeft",
           "GetPointDepthRight", "sad_CrossCheckDiaparity", "NCC_algorithem", "ncc_set_impl", "sad_set_impl", "asw_masks",
           "AdaptiveSupportWeight", "sad_batch", "ncc_batch", "asw_batch", "asw_set_impl", "asw_CrossCheckDiaparity", "cvtColor_BGR2GRAY", "copyMakeBorder_replicate",
           "to_float", "MedianFilter", "RemoveSpeckles", "imread", "imwrite", "ADCensusOption", "adcensus_option_aggregate", "Pipeline", "scratch_trim", "scratch_info"]


def current_stream_ptr(device=None):
    """torch's current stream on `device` (default: the current device) as a void*."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _on_tensor_device(f):
    """Stateless entry points launch on the current HIP device: make the device of the first tensor argument current
    for the call (a no-op on a one-GPU process), so that the kernels and the stream belong to the tensors' GPU."""
    import functools

    @functools.wraps(f)
    def g(*a, **kw):
        for t in list(a) + list(kw.values()):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                if t.device.index is not None and t.device.index != torch.cuda.current_device():
                    with torch.cuda.device(t.device):
                        return f(*a, **kw)
                break
        return f(*a, **kw)
    return g


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dev(t, dtype, shape=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a torch tensor on the GPU")
    if t.dtype != dtype or not t.is_contiguous():
        raise TypeError(f"{name} must be contiguous {dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _dev_index(device):
    """HIP device ordinal of a torch device (cuda without an index = the current one)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise TypeError(f"{device} is not a GPU")
    return torch.cuda.current_device() if device.index is None else int(device.index)


def _view_of(base_ptr, shape, dtype, device):
    """Zero-copy torch view of a library-owned device buffer (borrowed pointer)."""
    n = 1
    for s in shape:
        n *= s
    esz = torch.empty((), dtype=dtype).element_size()

    class _Holder:  # __cuda_array_interface__ provider
        pass

    h = _Holder()
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1", torch.float64: "<f8"}[dtype]
    h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(base_ptr), False),
                                  "version": 2, "strides": None}
    return torch.as_tensor(h, device=device)


class AD_Census:
    """class AD_Census (AD-CensusV1/AD-Census.h:9-43).

    Initialize(leftImage, rightImage, dispRange, row, col, LImage, RImage, sigmaC, sigmaS)
    -> here the images are bound at ComputeADcensus time and the two unused Mat
    arguments are dropped.
    """

    def __init__(self):
        self._h = None

    def Initialize(self, leftImage, rightImage, dispRange, row, col, sigmaC, sigmaS, placement_search=True, store_calibration=True):
        """AD-Census.h:322-344.  placement_search / store_calibration = False skip the two measuring steps of
        smt_adcensus_create (smt_adcensus_create_ex flags): for handles created per request."""
        self.row, self.col, self.dispRange = int(row), int(col), int(dispRange)
        self._L = _dev(leftImage, torch.float32, (row, col), "leftImage")
        self._R = _dev(rightImage, torch.float32, (row, col), "rightImage")
        self.device = leftImage.device
        h = C.c_void_p()
        if rightImage.device != self.device:
            raise ValueError("leftImage and rightImage live on different devices")
        flags = (0 if placement_search else 1) | (0 if store_calibration else 2)
        check(lib().smt_adcensus_create_ex(_dev_index(self.device), self.row, self.col, self.dispRange, C.c_float(sigmaC),
                                           C.c_float(sigmaS), C.c_uint(flags), C.byref(h)), "smt_adcensus_create_ex")
        self._h = h
        self._views = 0
        return self

    def _bind_stream(self):
        check(lib().smt_adcensus_set_stream(self._h, current_stream_ptr(self.device)), "smt_adcensus_set_stream")

    def _compute(self, views, dispL=None, dispR=None):
        self._bind_stream()
        check(lib().smt_adcensus_compute(self._h, _ptr(self._L), _ptr(self._R), views, _ptr(dispL),
                                         _ptr(dispR)), "smt_adcensus_compute")

    def ComputeADcensus(self):
        """AD-Census.h:271-294"""
        self._compute(VIEW_LEFT)

    def ComputeADcensusRight(self):
        """AD-Census.h:296-318"""
        self._compute(VIEW_RIGHT)

    def ComputeBoth(self, leftdisp=None, rightDisp=None):
        """ComputeADcensus + ComputeADcensusRight + WTA in one fused launch."""
        if leftdisp is not None:
            _dev(leftdisp, torch.float32, (self.row, self.col), "leftdisp")
            _dev(rightDisp, torch.float32, (self.row, self.col), "rightDisp")
        self._compute(VIEW_BOTH, leftdisp, rightDisp)

    def ComputeBatch(self, L, R, leftdisp, rightDisp, views=VIEW_BOTH):
        """[pairs][row][col] batches; volumes are reused per pair (smt_adcensus_compute_batch)."""
        pairs = L.shape[0]
        _dev(L, torch.float32, (pairs, self.row, self.col), "L")
        _dev(R, torch.float32, (pairs, self.row, self.col), "R")
        _dev(leftdisp, torch.float32, (pairs, self.row, self.col), "leftdisp")
        _dev(rightDisp, torch.float32, (pairs, self.row, self.col), "rightDisp")
        self._bind_stream()
        check(lib().smt_adcensus_compute_batch(self._h, _ptr(L), _ptr(R), pairs, views, _ptr(leftdisp),
                                               _ptr(rightDisp)), "smt_adcensus_compute_batch")

    def WTA(self, leftdisp, rightDisp):
        """AD-Census.h:346-380 over the volumes already computed."""
        _dev(leftdisp, torch.float32, (self.row, self.col), "leftdisp")
        _dev(rightDisp, torch.float32, (self.row, self.col), "rightDisp")
        wta(self.GetPtrLeft(), leftdisp)
        wta(self.GetPtrRight(), rightDisp)

    def _vol(self, view):
        p = C.c_void_p()
        check(lib().smt_adcensus_volume(self._h, view, C.byref(p)), "smt_adcensus_volume")
        return _view_of(p.value, (self.row, self.col, self.dispRange), torch.float32, self.device)

    def GetPtrLeft(self):
        """AD-Census.h:50-60 (borrowed view of costVolume)."""
        return self._vol(VIEW_LEFT)

    def GetPtrRight(self):
        """AD-Census.h:62-72"""
        return self._vol(VIEW_RIGHT)

    def status(self):
        check(lib().smt_adcensus_status(self._h), "smt_adcensus_status")

    def force_generic(self, on=True):
        check(lib().smt_adcensus_force_generic(self._h, int(on)), "smt_adcensus_force_generic")

    def timing(self, enable=True):
        """False / 0: off.  True / 1: record HIP events around the kernels of every pair; N > 1: of every
        N-th pair (an event record costs about 3 us on the stream)."""
        check(lib().smt_adcensus_timing(self._h, int(enable)), "smt_adcensus_timing")

    def kernel_times(self):
        """(prep_ms[], cost_ms[]) of the pairs processed since timing(True); synchronises."""
        cap = 1024
        a = (C.c_float * cap)()
        b = (C.c_float * cap)()
        n = C.c_int()
        check(lib().smt_adcensus_kernel_times(self._h, a, b, cap, C.byref(n)), "smt_adcensus_kernel_times")
        return list(a[:n.value]), list(b[:n.value])

    def placement(self):
        """(candidate volume pairs tried at Initialize, store-only ms of the pair kept)."""
        n, ms = C.c_int(), C.c_float()
        check(lib().smt_adcensus_placement(self._h, C.byref(n), C.byref(ms)), "smt_adcensus_placement")
        return n.value, ms.value

    def store_mode(self):
        """(plain stores chosen?, calibration ms with streaming stores, with plain stores)."""
        pl, a, b = C.c_int(), C.c_float(), C.c_float()
        check(lib().smt_adcensus_store_mode(self._h, C.byref(pl), C.byref(a), C.byref(b)), "smt_adcensus_store_mode")
        return bool(pl.value), a.value, b.value

    def diag(self, reps=20):
        """smt_adcensus_diag: (in-kernel shader clock in MHz, stamped cost-kernel ms, store-only ms)."""
        self._bind_stream()
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        check(lib().smt_adcensus_diag(self._h, int(reps), C.byref(a), C.byref(b), C.byref(c)), "smt_adcensus_diag")
        return a.value, b.value, c.value

    def close(self):
        if self._h is not None:
            lib().smt_adcensus_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@_on_tensor_device
def wta(vol, disp=None):
    """First-strict-minimum argmin over d (CrossArm.cpp:33-57, ScanlineOptimizer.h:40-64,
    CBLSM.h:383-407)."""
    H, W, D = vol.shape
    _dev(vol, torch.float32, name="vol")
    if disp is None:
        disp = torch.empty((H, W), dtype=torch.float32, device=vol.device)
    _dev(disp, torch.float32, (H, W), "disp")
    check(lib().smt_wta(_ptr(vol), H, W, D, _ptr(disp), current_stream_ptr()), "smt_wta")
    return disp


# ======================================================================================
# CrossArmAggregation  (AD-CensusV1/CrossArm.h:9-36) + CBLSM.h arms / costAggregationV5
# ======================================================================================
class CrossArmAggregation:
    """Initialize(row, col, leftImage, rightImage, tao, dispRange) -- the two float image
    pointers are stored but never used by the reference (CrossArm.cpp:10-11) and are
    dropped here.  `style="cblsm"` selects the CBLSM.cpp:28-32 constants (tau=25, by-value
    threshold, no right-arm stride bug)."""

    def __init__(self):
        self._h = None

    def Initialize(self, row, col, tao, dispRange, device=None, style="adcensus", quirks=None,
                   sec_length=17, max_length=34, tau_low=6):
        self.close()
        self.row, self.col, self.dispRange = int(row), int(col), int(dispRange)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        p = _lib.CrossArmParams()
        if style == "cblsm":
            lib().smt_crossarm_cblsm_params(C.byref(p))
        else:
            lib().smt_crossarm_default_params(C.byref(p))
        p.tau, p.tau_low, p.sec_length, p.max_length = int(tao), tau_low, sec_length, max_length
        if quirks is not None:
            p.quirks = quirks
        h = C.c_void_p()
        check(lib().smt_crossarm_create_on(_dev_index(self.device), self.row, self.col, self.dispRange, C.byref(p), C.byref(h)),
              "smt_crossarm_create_on")
        self._h = h
        return self

    def _bind(self):
        check(lib().smt_crossarm_set_stream(self._h, current_stream_ptr(self.device)), "smt_crossarm_set_stream")

    def ComputeArmLengths(self, Image):
        """ComputeLeftArmLength, ComputeRightArmLength, ComputeTopArmLength,
        ComputeButtonArmLength (CrossArm.cpp:147-598) in main.cpp:69-72's order.
        Image: uint8 [row][col] or [row][col][3]."""
        ch = 1 if Image.dim() == 2 else int(Image.shape[2])
        _dev(Image, torch.uint8, (self.row, self.col) if ch == 1 else (self.row, self.col, ch), "Image")
        self._bind()
        check(lib().smt_crossarm_arms(self._h, _ptr(Image), ch), "smt_crossarm_arms")

    def Reset(self):
        """The state part of Initialize (CrossArm.cpp:13-17): `_tao = tao`, four zeroed maps."""
        self._bind()
        check(lib().smt_crossarm_reset(self._h), "smt_crossarm_reset")

    def _arm_dir(self, Image, dirn):
        ch = 1 if Image.dim() == 2 else int(Image.shape[2])
        _dev(Image, torch.uint8, (self.row, self.col) if ch == 1 else (self.row, self.col, ch), "Image")
        self._bind()
        check(lib().smt_crossarm_arm_dir(self._h, _ptr(Image), ch, dirn), "smt_crossarm_arm_dir")

    def ComputeLeftArmLength(self, Image):
        """CrossArm.cpp:147-260, with the sticky threshold as the previous call left it."""
        self._arm_dir(Image, 0)

    def ComputeRightArmLength(self, Image):
        """CrossArm.cpp:262-373 (`col = _row`, :265)."""
        self._arm_dir(Image, 1)

    def ComputeTopArmLength(self, Image):
        """CrossArm.cpp:375-486."""
        self._arm_dir(Image, 2)

    def ComputeButtonArmLength(self, Image):
        """CrossArm.cpp:488-598."""
        self._arm_dir(Image, 3)

    def tao(self):
        """Current value of the member `_tao` (CrossArm.h:34)."""
        t = C.c_int()
        check(lib().smt_crossarm_tau(self._h, C.byref(t)), "smt_crossarm_tau")
        return t.value

    def arm_maps(self):
        ps = [C.c_void_p() for _ in range(4)]
        check(lib().smt_crossarm_arm_maps(self._h, *[C.byref(p) for p in ps]), "smt_crossarm_arm_maps")
        return [_view_of(p.value, (self.row, self.col), torch.int32, self.device) for p in ps]

    def load_arm_maps(self, left, right, top, bottom):
        """Arm maps computed elsewhere (int32 [row][col] device tensors) become this handle's maps -- the shape of
        costAggregationV5(dispvolume, CostVolume, ArmvolumeL, ArmvolumeR, ArmvolumeUp, ArmvolumeDown, ...)."""
        for t, nm in zip((left, right, top, bottom), ("left", "right", "top", "bottom")):
            _dev(t, torch.int32, (self.row, self.col), nm)
        self._bind()
        check(lib().smt_crossarm_load_arm_maps(self._h, _ptr(left), _ptr(right), _ptr(top), _ptr(bottom)),
              "smt_crossarm_load_arm_maps")

    def _agg(self, dispVolume, aggregatedCostVolume, order, disp):
        shp = (self.row, self.col, self.dispRange)
        _dev(dispVolume, torch.float32, shp, "dispVolume")
        _dev(aggregatedCostVolume, torch.float32, shp, "aggregatedCostVolume")
        if disp is not None:
            _dev(disp, torch.float32, (self.row, self.col), "disp")
        self._bind()
        check(lib().smt_crossarm_aggregate(self._h, _ptr(dispVolume), _ptr(aggregatedCostVolume), order,
                                           _ptr(disp)), "smt_crossarm_aggregate")

    def AggregationVertical(self, dispVolume, aggregatedCostVolume, disp=None):
        """CrossArm.cpp:60-102 (+ fused WTA :33-57 when disp is given)."""
        self._agg(dispVolume, aggregatedCostVolume, 0, disp)

    def Aggregation(self, dispVolume, aggregatedCostVolume, disp=None):
        """CrossArm.cpp:104-145 (no call site in the reference): rows outer, exclusive upper bounds; empty
        rectangles give NaN and status() raises SMT_ERR_REF_UB."""
        self._agg(dispVolume, aggregatedCostVolume, 2, disp)

    def costAggregationV5(self, dispvolume, CostVolume, disp=None):
        """CBLSM.h:1179-1224 (row-major add order)."""
        self._agg(dispvolume, CostVolume, 1, disp)

    def WTA(self, AggredCostVolume, disp):
        wta(AggredCostVolume, disp)

    def set_variant(self, variant):
        """12 = 4x4 pixels per wave sharing union taps, lock-step workgroups (default), 7 = the same with 2x8 tiles,
        6 = free-running, 4 / 5 / 3 = earlier shared-tap forms, 8-11 = flagged accumulate on the matrix pipe,
        0 = four pixels per wave, 1 plain walk, 2 pipelined walk (include/smt.h)."""
        check(lib().smt_crossarm_set_variant(self._h, int(variant)), "smt_crossarm_set_variant")

    def set_arm_walk(self, on=True):
        """Arms by the neighbour-by-neighbour kernels (independent formulation) instead of the bit-mask ones."""
        check(lib().smt_crossarm_set_arm_walk(self._h, int(on)), "smt_crossarm_set_arm_walk")

    def set_sweep(self, sweep):
        check(lib().smt_crossarm_set_sweep(self._h, int(sweep)), "smt_crossarm_set_sweep")

    def set_strip_width(self, w):
        check(lib().smt_crossarm_set_strip_width(self._h, int(w)), "smt_crossarm_set_strip_width")

    def status(self):
        check(lib().smt_crossarm_status(self._h), "smt_crossarm_status")

    def close(self):
        if getattr(self, "_h", None) is not None:
            lib().smt_crossarm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@_on_tensor_device
def cblsm_ComputeAD(L, R, dispRange, view=VIEW_LEFT, out=None):
    """CBLSM.h:327-353 (view left) / :355-381 (view right): uchar images -> float AD volume."""
    H, W = L.shape
    _dev(L, torch.uint8, (H, W), "L")
    _dev(R, torch.uint8, (H, W), "R")
    if out is None:
        out = torch.empty((H, W, dispRange), dtype=torch.float32, device=L.device)
    check(lib().smt_cblsm_ad(_ptr(L), _ptr(R), H, W, dispRange, view, _ptr(out), current_stream_ptr()),
          "smt_cblsm_ad")
    return out


@_on_tensor_device
def _choose_arm(dirn, own, vert, ArmRL, ArmRR, dispRange, Armvolume, row, col):
    for a in (own, vert, ArmRL, ArmRR):
        if a is not None:
            _dev(a, torch.int32, (row, col), "arm map")
    if Armvolume is None:
        Armvolume = torch.empty((row, col, dispRange), dtype=torch.int32, device=own.device)
    _dev(Armvolume, torch.int32, (row, col, dispRange), "Armvolume")
    check(lib().smt_cblsm_choose_arm_length(dirn, _ptr(own), _ptr(vert) if vert is not None else None, _ptr(ArmRL),
                                            _ptr(ArmRR), row, col, int(dispRange), _ptr(Armvolume),
                                            current_stream_ptr()), "smt_cblsm_choose_arm_length")
    return Armvolume


def chooseArmLengthLeft(ArmLL, ArmLR, ArmRL, ArmRR, dispRange, Armvolume, row, col):
    """CBLSM.h:65-102 (argument order of the reference; ArmLR is unused there too)."""
    return _choose_arm(0, ArmLL, None, ArmRL, ArmRR, dispRange, Armvolume, row, col)


def chooseArmLengthRight(ArmLL, ArmLR, ArmRL, ArmRR, dispRange, Armvolume, row, col):
    """CBLSM.h:104-147."""
    return _choose_arm(1, ArmLR, None, ArmRL, ArmRR, dispRange, Armvolume, row, col)


def chooseArmLengthUp(ArmLUp, ArmLDown, ArmRUp, ArmRDown, ArmRL, ArmRR, dispRange, Armvolume, row, col):
    """CBLSM.h:151-192."""
    return _choose_arm(2, ArmLUp, ArmRUp, ArmRL, ArmRR, dispRange, Armvolume, row, col)


def chooseArmLengthDown(ArmLUp, ArmLDown, ArmRUp, ArmRDown, ArmRL, ArmRR, dispRange, Armvolume, row, col):
    """CBLSM.h:195-236."""
    return _choose_arm(3, ArmLDown, ArmRDown, ArmRL, ArmRR, dispRange, Armvolume, row, col)


@_on_tensor_device
def costAggregationNew(leftImage, rightImage, CostVolume, ArmvolumeL, ArmvolumeR, ArmvolumeUp, ArmvolumeDown, dispRange,
                       _row_, _col_, winSize):
    """CBLSM.h:1087-1126 (argument order of the reference).  Padded uint8 images, int32 [row][col][D] arm
    volumes, float32 [row][col][D] CostVolume (allocated when None)."""
    w = winSize + 1
    _dev(leftImage, torch.uint8, (_row_ + 2 * w, _col_ + 2 * w), "leftImage")
    _dev(rightImage, torch.uint8, (_row_ + 2 * w, _col_ + 2 * w), "rightImage")
    for a in (ArmvolumeL, ArmvolumeR, ArmvolumeUp, ArmvolumeDown):
        _dev(a, torch.int32, (_row_, _col_, dispRange), "arm volume")
    if CostVolume is None:
        CostVolume = torch.empty((_row_, _col_, dispRange), dtype=torch.float32, device=leftImage.device)
    _dev(CostVolume, torch.float32, (_row_, _col_, dispRange), "CostVolume")
    check(lib().smt_cblsm_cost_aggregation_new(_ptr(leftImage), _ptr(rightImage), _row_, _col_, int(dispRange), int(winSize),
                                               _ptr(ArmvolumeL), _ptr(ArmvolumeR), _ptr(ArmvolumeUp), _ptr(ArmvolumeDown),
                                               _ptr(CostVolume), current_stream_ptr()), "smt_cblsm_cost_aggregation_new")
    return CostVolume


# ======================================================================================
# ScanlineOptimizer  (AD-CensusV1/ScanlineOptimizer.h:8-34)
# ======================================================================================
class ScanlineOptimizer:
    def __init__(self):
        self._h = None

    def Initialize(self, row, col, dispRange, p1, p2, device=None):
        """:66-79 (the costVolume pointer argument is re-passed to ScanLine anyway)."""
        self.close()
        self.row, self.col, self.dispRange = int(row), int(col), int(dispRange)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        h = C.c_void_p()
        check(lib().smt_scanline_create_on(_dev_index(self.device), self.row, self.col, self.dispRange, int(p1), int(p2), C.byref(h)),
              "smt_scanline_create_on")
        self._h = h
        self._processed = None
        return self

    def ScanLine(self, costVolume, Image, out=None, disp=None):
        """:104-128; returns `_ProcessedVolume`.  Image = float32 gray guidance."""
        shp = (self.row, self.col, self.dispRange)
        _dev(costVolume, torch.float32, shp, "costVolume")
        _dev(Image, torch.float32, (self.row, self.col), "Image")
        if out is None:
            out = torch.empty(shp, dtype=torch.float32, device=costVolume.device)
        _dev(out, torch.float32, shp, "out")
        check(lib().smt_scanline_set_stream(self._h, current_stream_ptr(self.device)), "smt_scanline_set_stream")
        check(lib().smt_scanline_run(self._h, _ptr(costVolume), _ptr(Image), _ptr(out), _ptr(disp)),
              "smt_scanline_run")
        self._processed = out
        return out

    def ScanPass(self, costVolume, Image, which):
        """One path volume: 'left' (ScanLineLeftRight isLeft=true, :130-192), 'right', 'up'
        (ScanLineUpDown isUp=true, :194-253), 'down'."""
        pass_id = {"left": 0, "right": 1, "up": 2, "down": 3}[which]
        out = torch.empty_like(costVolume)
        check(lib().smt_scanline_set_stream(self._h, current_stream_ptr(self.device)), "smt_scanline_set_stream")
        check(lib().smt_scanline_pass(self._h, _ptr(costVolume), _ptr(Image), pass_id, _ptr(out)),
              "smt_scanline_pass")
        return out

    def WTA(self, disp):
        """:40-64 over `_ProcessedVolume`."""
        wta(self._processed, disp)

    def close(self):
        if getattr(self, "_h", None) is not None:
            lib().smt_scanline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ======================================================================================
# LeftRightConsistency  (AD-CensusV1/PostProcessing.h:72-135)
# ======================================================================================
@_on_tensor_device
def LeftRightConsistency(col, row, gate, leftDisp, rightDisp, want_lists=False):
    """In place on leftDisp (+inf = invalid).  Returns (cls uint8 [row][col], n_occlusion,
    n_mismatch[, occlusions, mismatches]); the lists are (row, col) pairs in the
    reference's row-major order."""
    _dev(leftDisp, torch.float32, (row, col), "leftDisp")
    _dev(rightDisp, torch.float32, (row, col), "rightDisp")
    cls = torch.empty((row, col), dtype=torch.uint8, device=leftDisp.device)
    counts = torch.zeros(2, dtype=torch.int32, device=leftDisp.device)
    check(lib().smt_lrcheck(_ptr(leftDisp), _ptr(rightDisp), row, col, int(gate), _ptr(cls), _ptr(counts),
                            current_stream_ptr()), "smt_lrcheck")
    n = counts.cpu().tolist()
    if not want_lists:
        return cls, n[0], n[1]
    import numpy as np
    ch = np.ascontiguousarray(cls.cpu().numpy())
    occ = np.empty((row * col, 2), np.int32)
    mis = np.empty((row * col, 2), np.int32)
    no, nm = C.c_int(), C.c_int()
    check(lib().smt_lrcheck_lists(ch.ctypes.data_as(C.c_void_p), row, col, occ.ctypes.data_as(C.c_void_p),
                                  C.byref(no), mis.ctypes.data_as(C.c_void_p), C.byref(nm)), "smt_lrcheck_lists")
    return cls, n[0], n[1], occ[:no.value], mis[:nm.value]


@_on_tensor_device
def LeftAndRightConsistency(leftDisp, rightDisp, lastDisp, col, row, gate):
    """PostProcessing.h:10-70 (argument order of the reference; no call site there): out of place, lastDisp
    receives the kept disparities and 0 for rejected pixels.  Returns (cls, n_occlusion, n_mismatch)."""
    _dev(leftDisp, torch.float32, (row, col), "leftDisp")
    _dev(rightDisp, torch.float32, (row, col), "rightDisp")
    _dev(lastDisp, torch.float32, (row, col), "lastDisp")
    cls = torch.empty((row, col), dtype=torch.uint8, device=leftDisp.device)
    counts = torch.zeros(2, dtype=torch.int32, device=leftDisp.device)
    check(lib().smt_lrcheck_variant(_ptr(leftDisp), _ptr(rightDisp), _ptr(lastDisp), row, col, C.c_float(gate),
                                    _ptr(cls), _ptr(counts), current_stream_ptr()), "smt_lrcheck_variant")
    n = counts.cpu().tolist()
    return cls, n[0], n[1]


@_on_tensor_device
def FillTheHole(row, col, dispRange, dispLeft, occlusion, mismatch):
    """PostProcessing.h:156-248, in place on the device map dispLeft ([row][col] float32; the
    reference's internal width/height swap is reproduced).  occlusion / mismatch: (first, second)
    pairs ([n, 2] int arrays or lists of pairs) in list order.  Returns the mismatch list as the
    reference leaves it: the third pass's pixels when that pass ran, else the input list."""
    import numpy as np
    _dev(dispLeft, torch.float32, (row, col), "dispLeft")
    occ = np.ascontiguousarray(np.asarray(occlusion, np.int32).reshape(-1, 2))
    mis = np.ascontiguousarray(np.asarray(mismatch, np.int32).reshape(-1, 2))
    third = np.empty((row * col, 2), np.int32)
    nt = C.c_int(-1)
    check(lib().smt_fill_the_hole(_ptr(dispLeft), row, col, int(dispRange), occ.ctypes.data_as(C.c_void_p),
                                  len(occ), mis.ctypes.data_as(C.c_void_p), len(mis),
                                  third.ctypes.data_as(C.c_void_p), C.byref(nt), current_stream_ptr()),
          "smt_fill_the_hole")
    return third[:nt.value].copy() if nt.value >= 0 else mis


# ======================================================================================
# CrossAggregator  (CBLSM/cross_aggregator.h:27-113)
# ======================================================================================
class CrossAggregator:
    def __init__(self):
        self._h = None

    def Initialize(self, width, height, min_disparity, max_disparity, device=None):
        """:19-58; returns False where the reference does."""
        self.close()
        self.width, self.height = int(width), int(height)
        self.disp_range = int(max_disparity) - int(min_disparity)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        h = C.c_void_p()
        rc = lib().smt_crossagg_create_on(_dev_index(self.device), self.width, self.height, self.disp_range, C.byref(h))
        if rc == -1:
            return False
        check(rc, "smt_crossagg_create")
        self._h = h
        return True

    def SetData(self, img_left, img_right, cost_init):
        """:60-65 (img_right is stored but never read by the reference)."""
        self._img = _dev(img_left, torch.uint8, (self.height, self.width, 3), "img_left")
        self._cost = _dev(cost_init, torch.float32, (self.height, self.width, self.disp_range), "cost_init")

    def SetParams(self, cross_L1, cross_L2, cross_t1, cross_t2):
        check(lib().smt_crossagg_set_params(self._h, int(cross_L1), int(cross_L2), int(cross_t1), int(cross_t2)),
              "smt_crossagg_set_params")

    def Aggregate(self, num_iters):
        """:89-118; silently does nothing when uninitialised, like the reference (:91-93)."""
        if self._h is None:
            return
        check(lib().smt_crossagg_set_stream(self._h, current_stream_ptr(self.device)), "smt_crossagg_set_stream")
        check(lib().smt_crossagg_aggregate(self._h, _ptr(self._img), _ptr(self._cost), int(num_iters)),
              "smt_crossagg_aggregate")

    def set_impl(self, impl):
        """2 = shared-tap passes (default), 1 = one pixel per wave (test hook)."""
        check(lib().smt_crossagg_set_impl(self._h, int(impl)), "smt_crossagg_set_impl")

    def get_cost_ptr(self):
        p = C.c_void_p()
        check(lib().smt_crossagg_cost(self._h, C.byref(p)), "smt_crossagg_cost")
        return _view_of(p.value, (self.height, self.width, self.disp_range), torch.float32, self.device)

    def get_arms_ptr(self):
        p = C.c_void_p()
        check(lib().smt_crossagg_arms(self._h, C.byref(p)), "smt_crossagg_arms")
        return _view_of(p.value, (self.height, self.width, 4), torch.uint8, self.device)

    def close(self):
        if getattr(self, "_h", None) is not None:
            lib().smt_crossagg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ======================================================================================
# Window matchers  (SAD/Sad.h, NCC/NCC.h, ASW/ASW.h)
# ======================================================================================
def GetPointDepthLeft(leftimg, rightimg, MaxDisparity, winsize):
    """Sad.h:96-139.  leftimg/rightimg: uint8 replicate-padded by winsize+1."""
    return _sad(leftimg, rightimg, MaxDisparity, winsize, VIEW_LEFT)


def GetPointDepthRight(leftimg, rightimg, MaxDisparity, winsize):
    """Sad.h:141-182."""
    return _sad(leftimg, rightimg, MaxDisparity, winsize, VIEW_RIGHT)


@_on_tensor_device
def _sad(Lp, Rp, D, winsize, view):
    w = winsize + 1
    Hp, Wp = Lp.shape
    H, W = Hp - 2 * w, Wp - 2 * w
    _dev(Lp, torch.uint8, (Hp, Wp), "leftimg")
    _dev(Rp, torch.uint8, (Hp, Wp), "rightimg")
    disp = torch.empty((H, W), dtype=torch.int32, device=Lp.device)
    check(lib().smt_sad(_ptr(Lp), _ptr(Rp), H, W, D, winsize, view, _ptr(disp), current_stream_ptr()), "smt_sad")
    return disp


@_on_tensor_device
def sad_CrossCheckDiaparity(leftdisp, rightdisp):
    """Sad.h:184-222 -> (lastdisp int32, cls uint8)."""
    H, W = leftdisp.shape
    _dev(leftdisp, torch.int32, (H, W), "leftdisp")
    _dev(rightdisp, torch.int32, (H, W), "rightdisp")
    out = torch.empty_like(leftdisp)
    cls = torch.empty((H, W), dtype=torch.uint8, device=leftdisp.device)
    check(lib().smt_sad_crosscheck(_ptr(leftdisp), _ptr(rightdisp), H, W, _ptr(out), _ptr(cls),
                                   current_stream_ptr()), "smt_sad_crosscheck")
    return out, cls


@_on_tensor_device
def NCC_algorithem(leftImage, rigthImage, winSize, dispRange, want_cost=False):
    """NCC.h:69-95.  uint8 [H][W] unpadded images -> int32 disparity (argmax)."""
    H, W = leftImage.shape
    _dev(leftImage, torch.uint8, (H, W), "leftImage")
    _dev(rigthImage, torch.uint8, (H, W), "rigthImage")
    disp = torch.empty((H, W), dtype=torch.int32, device=leftImage.device)
    cost = torch.full((H, W, dispRange), float("nan"), dtype=torch.float64, device=leftImage.device) if want_cost else None
    check(lib().smt_ncc(_ptr(leftImage), _ptr(rigthImage), H, W, dispRange, winSize, _ptr(disp), _ptr(cost),
                        current_stream_ptr()), "smt_ncc")
    return (disp, cost) if want_cost else disp


def sad_set_impl(impl):
    """2 = LDS-staged SAD kernel (default), 1 = first formulation (test hook)."""
    check(lib().smt_sad_set_impl(int(impl)), "smt_sad_set_impl")


def ncc_set_impl(impl):
    """2 = window statistics + dot4 cross term (default), 1 = the reference's loop nest (test hook)."""
    check(lib().smt_ncc_set_impl(int(impl)), "smt_ncc_set_impl")


def asw_masks(winSize, spaceSigma, colorSigma, device):
    """getGausssianMask (ASW.h:16-35) + getColorMask (:41-47), host float64 -> device tensors."""
    import numpy as np
    side = 2 * winSize + 3
    sp = np.empty((side, side), np.float64)
    cm = np.empty(256, np.float64)
    check(lib().smt_asw_masks(winSize, C.c_double(spaceSigma), C.c_double(colorSigma),
                              sp.ctypes.data_as(C.c_void_p), cm.ctypes.data_as(C.c_void_p)), "smt_asw_masks")
    return torch.from_numpy(sp).to(device), torch.from_numpy(cm).to(device)


@_on_tensor_device
def AdaptiveSupportWeight(leftGray, rightGray, winSize, dispRange, space, color, T, view=VIEW_LEFT, want_cost=False):
    """ASW.h:329-378 (view left) / :382-431 (view right).  Padded uint8 images."""
    wins = winSize + 1
    Hp, Wp = leftGray.shape
    H, W = Hp - 2 * wins, Wp - 2 * wins
    _dev(leftGray, torch.uint8, (Hp, Wp), "leftGray")
    _dev(rightGray, torch.uint8, (Hp, Wp), "rightGray")
    _dev(space, torch.float64, (2 * winSize + 3, 2 * winSize + 3), "space")
    _dev(color, torch.float64, (256,), "color")
    disp = torch.empty((H, W), dtype=torch.float32, device=leftGray.device)
    cost = torch.empty((H, W, dispRange), dtype=torch.float32, device=leftGray.device) if want_cost else None
    check(lib().smt_asw(_ptr(leftGray), _ptr(rightGray), H, W, dispRange, winSize, _ptr(space), _ptr(color), int(T),
                        view, _ptr(disp), _ptr(cost), current_stream_ptr()), "smt_asw")
    return (disp, cost) if want_cost else disp


@_on_tensor_device
def sad_batch(leftimgs, rightimgs, MaxDisparity, winsize, view=VIEW_LEFT):
    """smt_sad_batch: [P, H+2w, W+2w] uint8 padded pairs -> int32 [P, H, W] (GetPointDepthLeft / Right per pair)."""
    w = winsize + 1
    P, Hp, Wp = leftimgs.shape
    H, W = Hp - 2 * w, Wp - 2 * w
    _dev(leftimgs, torch.uint8, (P, Hp, Wp), "leftimgs")
    _dev(rightimgs, torch.uint8, (P, Hp, Wp), "rightimgs")
    disp = torch.empty((P, H, W), dtype=torch.int32, device=leftimgs.device)
    check(lib().smt_sad_batch(_ptr(leftimgs), _ptr(rightimgs), P, C.c_size_t(0), H, W, MaxDisparity, winsize, view,
                              _ptr(disp), C.c_size_t(0), current_stream_ptr()), "smt_sad_batch")
    return disp


@_on_tensor_device
def ncc_batch(leftImages, rightImages, winSize, dispRange):
    """smt_ncc_batch: [P, H, W] uint8 pairs -> int32 [P, H, W] (NCC_algorithem per pair)."""
    P, H, W = leftImages.shape
    _dev(leftImages, torch.uint8, (P, H, W), "leftImages")
    _dev(rightImages, torch.uint8, (P, H, W), "rightImages")
    disp = torch.empty((P, H, W), dtype=torch.int32, device=leftImages.device)
    check(lib().smt_ncc_batch(_ptr(leftImages), _ptr(rightImages), P, C.c_size_t(0), H, W, dispRange, winSize, _ptr(disp),
                              C.c_size_t(0), current_stream_ptr()), "smt_ncc_batch")
    return disp


@_on_tensor_device
def asw_batch(leftGrays, rightGrays, winSize, dispRange, space, color, T, view=VIEW_LEFT):
    """smt_asw_batch: [P, H+2w, W+2w] uint8 padded pairs -> float32 [P, H, W] (AdaptiveSupportWeight per pair)."""
    wins = winSize + 1
    P, Hp, Wp = leftGrays.shape
    H, W = Hp - 2 * wins, Wp - 2 * wins
    _dev(leftGrays, torch.uint8, (P, Hp, Wp), "leftGrays")
    _dev(rightGrays, torch.uint8, (P, Hp, Wp), "rightGrays")
    _dev(space, torch.float64, (2 * winSize + 3, 2 * winSize + 3), "space")
    _dev(color, torch.float64, (256,), "color")
    disp = torch.empty((P, H, W), dtype=torch.float32, device=leftGrays.device)
    check(lib().smt_asw_batch(_ptr(leftGrays), _ptr(rightGrays), P, C.c_size_t(0), H, W, dispRange, winSize, _ptr(space),
                              _ptr(color), int(T), view, _ptr(disp), C.c_size_t(0), current_stream_ptr()), "smt_asw_batch")
    return disp


def asw_set_impl(impl):
    """0 = automatic (default: 3 up to a 6 GiB table, 6 beyond), 3 / 4 / 5 = whole-image anchor table variants, 6 = per-workgroup anchor slots, 1 = first formulation (test hook)."""
    check(lib().smt_asw_set_impl(int(impl)), "smt_asw_set_impl")


@_on_tensor_device
def asw_CrossCheckDiaparity(leftdisp, rightdisp):
    """ASW.h:108-145 -> uint8 map (0 = rejected)."""
    H, W = leftdisp.shape
    _dev(leftdisp, torch.float32, (H, W), "leftdisp")
    _dev(rightdisp, torch.float32, (H, W), "rightdisp")
    out = torch.empty((H, W), dtype=torch.uint8, device=leftdisp.device)
    check(lib().smt_asw_crosscheck(_ptr(leftdisp), _ptr(rightdisp), H, W, _ptr(out), current_stream_ptr()),
          "smt_asw_crosscheck")
    return out


# ======================================================================================
# Either side of the path: input staging and the first post-filter
# ======================================================================================
@_on_tensor_device
def cvtColor_BGR2GRAY(bgr):
    """cvtColor(img, gray, CV_BGR2GRAY) (main.cpp:19-20); uint8 [H][W][3] -> uint8 [H][W]."""
    H, W, _ = bgr.shape
    _dev(bgr, torch.uint8, (H, W, 3), "bgr")
    gray = torch.empty((H, W), dtype=torch.uint8, device=bgr.device)
    check(lib().smt_bgr2gray(_ptr(bgr), H, W, _ptr(gray), current_stream_ptr()), "smt_bgr2gray")
    return gray


@_on_tensor_device
def copyMakeBorder_replicate(img, pad):
    """copyMakeBorder(img, out, pad, pad, pad, pad, BORDER_REPLICATE) (SADmain.cpp:47-48)."""
    H, W = img.shape
    _dev(img, torch.uint8, (H, W), "img")
    out = torch.empty((H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=img.device)
    check(lib().smt_pad_replicate(_ptr(img), H, W, int(pad), _ptr(out), current_stream_ptr()), "smt_pad_replicate")
    return out


@_on_tensor_device
def to_float(img):
    """uchar -> float copy of main.cpp:46-55."""
    H, W = img.shape
    _dev(img, torch.uint8, (H, W), "img")
    out = torch.empty((H, W), dtype=torch.float32, device=img.device)
    check(lib().smt_u8_to_f32(_ptr(img), H, W, _ptr(out), current_stream_ptr()), "smt_u8_to_f32")
    return out


@_on_tensor_device
def MedianFilter(inp, width, height, wnd_size):
    """PostProcessing.h:314-344."""
    _dev(inp, torch.float32, (height, width), "in")
    out = torch.empty_like(inp)
    check(lib().smt_median_filter(_ptr(inp), _ptr(out), width, height, int(wnd_size), current_stream_ptr()),
          "smt_median_filter")
    return out


@_on_tensor_device
def RemoveSpeckles(disparity_map, width, height, diff_insame, min_speckle_aera, invalid_val):
    """PostProcessing.h:250-311, in place.  invalid_val: int, as in the reference's signature."""
    _dev(disparity_map, torch.float32, (height, width), "disparity_map")
    check(lib().smt_remove_speckles(_ptr(disparity_map), width, height, int(diff_insame), C.c_uint(min_speckle_aera),
                                    int(invalid_val), current_stream_ptr()), "smt_remove_speckles")
    return disparity_map


# ======================================================================================
# Image files (host side): imread / imwrite of the reference's drivers
# ======================================================================================
def imread(path, want_channels=3):
    """cv::imread(path) (main.cpp:16-17): numpy uint8 [H][W][3] in B, G, R order by default;
    want_channels=1 gray, 0 as stored.  PNG / PGM / PPM, decoded by libsmt_hip.so's own reader."""
    import numpy as np
    p = C.POINTER(C.c_uint8)()
    H, W, ch = C.c_int(), C.c_int(), C.c_int()
    check(lib().smt_image_read(str(path).encode(), int(want_channels), C.byref(p), C.byref(H), C.byref(W), C.byref(ch)),
          "smt_image_read")
    try:
        a = np.ctypeslib.as_array(p, shape=(H.value * W.value * ch.value,)).copy()
    finally:
        lib().smt_image_free(p)
    return a.reshape(H.value, W.value) if ch.value == 1 else a.reshape(H.value, W.value, ch.value)


def imwrite(path, img):
    """cv::imwrite(path, img) (main.cpp:115-117) for uint8 [H][W] or [H][W][3] (B, G, R) arrays /
    CPU tensors; .png, .pgm, .ppm."""
    import numpy as np
    a = np.ascontiguousarray(img.cpu().numpy() if isinstance(img, torch.Tensor) else img, dtype=np.uint8)
    H, W = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    check(lib().smt_image_write(str(path).encode(), a.ctypes.data_as(C.c_void_p), H, W, ch), "smt_image_write")


# ======================================================================================
# ADCensusOption (CBLSM/adcensus_types.h:45-75) and the caller shape of CBLSM.cpp:138-143
# ======================================================================================
def ADCensusOption(**overrides):
    """Default-constructed ADCensusOption (adcensus_types.h:69-75) with fields overridden by keyword."""
    o = _lib.ADCensusOption()
    lib().smt_adcensus_option_default(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


@_on_tensor_device
def adcensus_option_aggregate(option, bytes_left, dispVolum, num_iters=4, want_disp=True):
    """CBLSM.cpp:138-143 + :152: CrossAggregator driven by an ADCensusOption -> (cost, disp)."""
    H, W, _ = bytes_left.shape
    D = option.max_disparity - option.min_disparity
    _dev(bytes_left, torch.uint8, (H, W, 3), "bytes_left")
    _dev(dispVolum, torch.float32, (H, W, D), "dispVolum")
    cost = torch.empty((H, W, D), dtype=torch.float32, device=dispVolum.device)
    disp = torch.empty((H, W), dtype=torch.float32, device=dispVolum.device) if want_disp else None
    check(lib().smt_adcensus_option_aggregate(C.byref(option), _ptr(bytes_left), _ptr(dispVolum), W, H, int(num_iters),
                                              _ptr(cost), _ptr(disp), current_stream_ptr()), "smt_adcensus_option_aggregate")
    return cost, disp


# ======================================================================================
# The whole AD-CensusV1/main.cpp pipeline, batched (smt_pipeline_*)
# ======================================================================================
def scratch_trim(keep_bytes=0, device=None):
    """smt_scratch_trim on `device` (default: current): idle scratch of smt_asw / smt_ncc back to the driver."""
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib().smt_scratch_trim(C.c_size_t(int(keep_bytes))), "smt_scratch_trim")


def scratch_info(device=None):
    """(reserved_bytes, used_bytes) of the library's scratch arena on `device`."""
    r, u = C.c_size_t(0), C.c_size_t(0)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib().smt_scratch_info(C.byref(r), C.byref(u)), "smt_scratch_info")
    return int(r.value), int(u.value)


class Pipeline:
    """main.cpp:46-92 (scanline and LR check enabled) for batches of gray pairs; the sharding unit of config 3."""

    def __init__(self, row, col, dispRange, device=None, **params):
        self.row, self.col, self.dispRange = int(row), int(col), int(dispRange)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        p = _lib.PipelineParams()
        lib().smt_pipeline_default_params(C.byref(p))
        for k, v in params.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        h = C.c_void_p()
        check(lib().smt_pipeline_create_on(_dev_index(self.device), self.row, self.col, self.dispRange, C.byref(p), C.byref(h)),
              "smt_pipeline_create_on")
        self._h = h

    def run(self, grayL, grayR):
        """uint8 [pairs][row][col] (or [row][col]) -> (dispL after LR check, dispR, cls, counts[pairs][2])."""
        if grayL.dim() == 2:
            grayL, grayR = grayL[None], grayR[None]
        P = grayL.shape[0]
        if _dev_index(grayL.device) != _dev_index(self.device) or grayR.device != grayL.device:
            raise ValueError(f"pipeline handle lives on {self.device}, images on {grayL.device} / {grayR.device}")
        _dev(grayL, torch.uint8, (P, self.row, self.col), "grayL")
        _dev(grayR, torch.uint8, (P, self.row, self.col), "grayR")
        dl = torch.empty((P, self.row, self.col), dtype=torch.float32, device=grayL.device)
        dr = torch.empty_like(dl)
        cls = torch.empty((P, self.row, self.col), dtype=torch.uint8, device=grayL.device)
        counts = torch.zeros((P, 2), dtype=torch.int32, device=grayL.device)
        check(lib().smt_pipeline_set_stream(self._h, current_stream_ptr(self.device)), "smt_pipeline_set_stream")
        check(lib().smt_pipeline_run_batch(self._h, _ptr(grayL), _ptr(grayR), P, _ptr(dl), _ptr(dr), _ptr(cls), _ptr(counts)),
              "smt_pipeline_run_batch")
        return dl, dr, cls, counts

    def volumes(self):
        ps = [C.c_void_p() for _ in range(5)]
        check(lib().smt_pipeline_volumes(self._h, *[C.byref(p) for p in ps]), "smt_pipeline_volumes")
        shp = (self.row, self.col, self.dispRange)
        return [_view_of(p.value, shp, torch.float32, self.device) for p in ps]

    def status(self):
        check(lib().smt_pipeline_status(self._h), "smt_pipeline_status")

    def close(self):
        if getattr(self, "_h", None) is not None:
            lib().smt_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

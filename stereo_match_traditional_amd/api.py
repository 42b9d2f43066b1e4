"""Host-side mirrors of the reference's classes / free functions over the C ABI.

Names and argument meaning follow the reference (file:line cited per class); buffers are
torch CUDA(=HIP) tensors whose data_ptr() is handed to libsmt_hip.so.  Nothing here
computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, VIEW_LEFT, VIEW_RIGHT, VIEW_BOTH

__all__ = ["AD_Census", "wta", "current_stream_ptr"]


def current_stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


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

    def Initialize(self, leftImage, rightImage, dispRange, row, col, sigmaC, sigmaS):
        self.row, self.col, self.dispRange = int(row), int(col), int(dispRange)
        self._L = _dev(leftImage, torch.float32, (row, col), "leftImage")
        self._R = _dev(rightImage, torch.float32, (row, col), "rightImage")
        self.device = leftImage.device
        h = C.c_void_p()
        check(lib().smt_adcensus_create(self.row, self.col, self.dispRange, C.c_float(sigmaC),
                                        C.c_float(sigmaS), C.byref(h)), "smt_adcensus_create")
        self._h = h
        self._views = 0
        return self

    def _bind_stream(self):
        check(lib().smt_adcensus_set_stream(self._h, current_stream_ptr()), "smt_adcensus_set_stream")

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
        check(lib().smt_adcensus_timing(self._h, int(enable)), "smt_adcensus_timing")

    def kernel_times(self):
        """(prep_ms[], cost_ms[]) of the pairs processed since timing(True); synchronises."""
        cap = 1024
        a = (C.c_float * cap)()
        b = (C.c_float * cap)()
        n = C.c_int()
        check(lib().smt_adcensus_kernel_times(self._h, a, b, cap, C.byref(n)), "smt_adcensus_kernel_times")
        return list(a[:n.value]), list(b[:n.value])

    def close(self):
        if self._h is not None:
            lib().smt_adcensus_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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

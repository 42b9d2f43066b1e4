"""ctypes binding of libsmt_hip.so (the C ABI in include/smt.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded the import of
any compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMT_HIP_LIB") or os.path.join(_HERE, "lib", "libsmt_hip.so")  # override: A/B kernel builds

SMT_OK = 0
SMT_ERR_DOMAIN = -4
SMT_ERR_REF_UB = -5
VIEW_LEFT, VIEW_RIGHT, VIEW_BOTH = 1, 2, 3
QUIRK_FIX_RIGHT_ARM_STRIDE = 0x1


class SmtError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(f"{what}: {strerror(status)} (status {status}, hip {last_hip_error()})")


class CrossArmParams(C.Structure):
    _fields_ = [("tau", C.c_int), ("tau_low", C.c_int), ("sec_length", C.c_int),
                ("max_length", C.c_int), ("chain_tau", C.c_int), ("quirks", C.c_uint)]


class PipelineParams(C.Structure):
    _fields_ = [("sigmaC", C.c_float), ("sigmaS", C.c_float), ("tao", C.c_int), ("p1", C.c_int), ("p2", C.c_int),
                ("gate", C.c_int)]


class ADCensusOption(C.Structure):
    """struct ADCensusOption (CBLSM/adcensus_types.h:45-75)."""
    _fields_ = [("min_disparity", C.c_int32), ("max_disparity", C.c_int32), ("lambda_ad", C.c_int32),
                ("lambda_census", C.c_int32), ("cross_L1", C.c_int32), ("cross_L2", C.c_int32), ("cross_t1", C.c_int32),
                ("cross_t2", C.c_int32), ("so_p1", C.c_float), ("so_p2", C.c_float), ("so_tso", C.c_int32),
                ("irv_ts", C.c_int32), ("irv_th", C.c_float), ("lrcheck_thres", C.c_float), ("do_lr_check", C.c_int32),
                ("do_filling", C.c_int32), ("do_discontinuity_adjustment", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m stereo_match_traditional_amd.build` "
                "(hipcc, gfx950).  This package has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.smt_strerror.restype = C.c_char_p
    return _lib


def strerror(status):
    return lib().smt_strerror(int(status)).decode()


def last_hip_error():
    return int(lib().smt_last_hip_error())


def check(status, what):
    if status != SMT_OK:
        raise SmtError(status, what)

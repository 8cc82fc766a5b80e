"""ctypes binding of the synthetic FM generator (tools/libfmsig.so). Test / bench infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FmsigParams(C.Structure):
    _fields_ = [
        ("fs", C.c_double),
        ("f_offset", C.c_double),
        ("dev", C.c_double),
        ("amp", C.c_double),
        ("a_mono", C.c_double),
        ("a_stereo", C.c_double),
        ("a_pilot", C.c_double),
        ("a_rds", C.c_double),
        ("f_left", C.c_double),
        ("f_right", C.c_double),
        ("noise_sigma", C.c_double),
        ("seed", C.c_uint64),
        ("pi", C.c_uint16),
        ("pty", C.c_uint8),
        ("ms", C.c_uint8),
        ("ps", C.c_char * 8),
    ]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfmsig.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "..", "oracle"),
                                   "../tools/libfmsig.so"])
        L = C.CDLL(path)
        L.fmsig_default.argtypes = [C.POINTER(FmsigParams), C.c_double]
        L.fmsig_generate_u8.argtypes = [C.POINTER(FmsigParams), C.c_uint64, C.c_uint32, C.c_void_p]
        L.fmsig_generate_f32.argtypes = [C.POINTER(FmsigParams), C.c_uint64, C.c_uint32, C.c_void_p]
        L.fmsig_rds_dbits.argtypes = [C.POINTER(FmsigParams), C.c_void_p]
        L.fmsig_rds_groups.argtypes = [C.POINTER(FmsigParams), C.c_void_p]
        L.fmsig_u8_to_f32.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        _LIB = L
    return _LIB


def default_params(fs, **kw):
    p = FmsigParams()
    lib().fmsig_default(C.byref(p), fs)
    for k, v in kw.items():
        if k == "ps":
            v = v.encode("latin1") if isinstance(v, str) else v
            v = v.ljust(8)[:8]
        setattr(p, k, v)
    return p


def mono_params(fs, **kw):
    """config-1 style station: L == R, no pilot, no 38 kHz term, no RDS."""
    base = dict(a_stereo=0.0, a_pilot=0.0, a_rds=0.0, f_right=1000.0, f_left=1000.0)
    base.update(kw)
    return default_params(fs, **base)


def generate_u8(p, start, n):
    out = np.empty(2 * n, dtype=np.uint8)
    lib().fmsig_generate_u8(C.byref(p), start, n, out.ctypes.data)
    return out


def generate_f32(p, start, n):
    out = np.empty(2 * n, dtype=np.float32)
    lib().fmsig_generate_f32(C.byref(p), start, n, out.ctypes.data)
    return out


def u8_to_f32(u8):
    u8 = np.ascontiguousarray(u8, dtype=np.uint8)
    out = np.empty(u8.size, dtype=np.float32)
    lib().fmsig_u8_to_f32(u8.ctypes.data, u8.size, out.ctypes.data)
    return out


def rds_dbits(p):
    out = np.empty(832, dtype=np.uint8)
    lib().fmsig_rds_dbits(C.byref(p), out.ctypes.data)
    return out


def rds_groups(p):
    out = np.empty((4, 4), dtype=np.uint16)
    lib().fmsig_rds_groups(C.byref(p), out.ctypes.data)
    return out

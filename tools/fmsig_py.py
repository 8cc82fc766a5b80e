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


# ---- device generator (libfmsig_hip.so, HIP) ---------------------------------------------------
_DLIB = None
CHAN_DTYPE = np.dtype([
    ("inv_fs", "<f8"), ("f_offset", "<f8"), ("dev", "<f8"), ("amp", "<f8"),
    ("a_mono", "<f8"), ("a_stereo", "<f8"), ("a_pilot", "<f8"), ("a_rds", "<f8"),
    ("f_left", "<f8"), ("f_right", "<f8"), ("noise_sigma", "<f8"), ("seed", "<u8"),
])


def dlib():
    global _DLIB
    if _DLIB is None:
        import torch  # noqa: F401  (one HIP runtime per process, see the product package)
        path = os.path.join(_HERE, "..", "pvr.rtl.radiofm_amd", "libfmsig_hip.so")
        L = C.CDLL(path)
        L.fmsig_device_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint64,
                                            C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
        L.fmsig_device_generate_u8.argtypes = L.fmsig_device_generate.argtypes
        L.fmsig_chan_size.restype = C.c_uint
        assert L.fmsig_chan_size() == CHAN_DTYPE.itemsize
        _DLIB = L
    return _DLIB


def channel_params(fs, channel, base_seed=1000, noise_sigma=0.01):
    """Station `channel` of the synthetic multi-channel workload (config 3/4): its own audio
    tones, noise seed, PI code and PS name."""
    return default_params(
        fs, noise_sigma=noise_sigma, seed=base_seed + channel,
        f_left=400.0 + 13.0 * (channel % 97), f_right=2500.0 + 7.0 * (channel % 211),
        pi=0x1000 + (channel % 0xE000),
        ps="C%07d" % (channel % 10000000))


class DeviceGenerator:
    """Generates [C][n] complex64 IQ on the GPU for a list of FmsigParams."""

    def __init__(self, params_list, device="cuda"):
        import torch
        self.torch = torch
        Cn = len(params_list)
        chans = np.zeros(Cn, dtype=CHAN_DTYPE)
        dbits = np.zeros((Cn, 832), dtype=np.uint8)
        for i, p in enumerate(params_list):
            chans[i] = (1.0 / p.fs, p.f_offset, p.dev, p.amp, p.a_mono, p.a_stereo, p.a_pilot,
                        p.a_rds, p.f_left, p.f_right, p.noise_sigma, p.seed)
            dbits[i] = rds_dbits(p)
        self.C = Cn
        self.d_chans = torch.from_numpy(chans.view(np.uint8).copy()).to(device)
        self.d_dbits = torch.from_numpy(dbits.reshape(-1).copy()).to(device)

    def generate(self, out, start, n):
        """out: torch cuda tensor, float32 viewable as [C, n, 2] (converted like the reference) or
        uint8 viewable as [C, n, 2] (raw RTL-SDR bytes); start: absolute sample index."""
        torch = self.torch
        assert out.is_cuda and out.dtype in (torch.float32, torch.uint8)
        assert out.numel() >= self.C * n * 2
        u8 = out.dtype == torch.uint8
        fn = dlib().fmsig_device_generate_u8 if u8 else dlib().fmsig_device_generate
        esz = 2 if u8 else 8
        stream = torch.cuda.current_stream().cuda_stream
        # grid.y is limited to 65535 channels per launch
        done = 0
        while done < self.C:
            cnt = min(32768, self.C - done)
            rc = fn(self.d_chans.data_ptr() + done * CHAN_DTYPE.itemsize,
                    self.d_dbits.data_ptr() + done * 832, 832, cnt, start, n,
                    out.data_ptr() + done * n * esz, n, stream)
            assert rc == 0
            done += cnt

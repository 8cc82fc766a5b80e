"""ctypes binding of the CPU oracle (oracle/libfmd_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FmoParams(C.Structure):
    _fields_ = [
        ("sample_rate_if", C.c_double),
        ("tuning_offset", C.c_double),
        ("sample_rate_pcm", C.c_double),
        ("bandwidth_pcm", C.c_double),
        ("downsample", C.c_uint),
        ("us_version", C.c_int),
        ("table_size", C.c_uint),
        ("if_filter_order", C.c_uint),
        ("tuning_shift_override", C.c_int),
        ("use_shift_override", C.c_int),
    ]


class FmoTaps(C.Structure):
    _fields_ = [
        ("n_demod", C.c_uint),
        ("demod", C.POINTER(C.c_float)),
        ("baseband", C.POINTER(C.c_float)),
        ("pilot38", C.POINTER(C.c_float)),
        ("n_audio", C.c_uint),
        ("mono_rs", C.POINTER(C.c_float)),
        ("stereo_rs", C.POINTER(C.c_float)),
        ("n_rds", C.c_uint),
        ("rds_lpf", C.POINTER(C.c_float)),
        ("rds_pll", C.POINTER(C.c_float)),
        ("rds_mf", C.POINTER(C.c_float)),
        ("rds_sync", C.POINTER(C.c_float)),
    ]


class FmoStatus(C.Structure):
    _fields_ = [
        ("stereo", C.c_int),
        ("tuning_offset", C.c_float),
        ("if_level", C.c_float),
        ("baseband_level", C.c_float),
        ("pilot_level", C.c_float),
        ("rds_state", C.c_int),
    ]


class FmoPacket(C.Structure):
    _fields_ = [("stream_id", C.c_int), ("size", C.c_int), ("pts", C.c_double),
                ("duration", C.c_double), ("data", C.POINTER(C.c_uint8))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfmd_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.fmo_create.restype = C.c_void_p
        L.fmo_create.argtypes = [C.POINTER(FmoParams)]
        L.fmo_destroy.argtypes = [C.c_void_p]
        L.fmo_reset.argtypes = [C.c_void_p]
        L.fmo_process_stream.restype = C.c_uint
        L.fmo_process_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        L.fmo_get_status.argtypes = [C.c_void_p, C.POINTER(FmoStatus)]
        L.fmo_get_taps.argtypes = [C.c_void_p, C.POINTER(FmoTaps)]
        L.fmo_rds_group_count.restype = C.c_uint
        L.fmo_rds_group_count.argtypes = [C.c_void_p]
        L.fmo_rds_group_get.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.POINTER(C.c_uint)]
        L.fmo_uecp_frame_count.restype = C.c_uint
        L.fmo_uecp_frame_count.argtypes = [C.c_void_p]
        L.fmo_uecp_frame_get.restype = C.c_uint
        L.fmo_uecp_frame_get.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
        L.fmo_channel_name.restype = C.c_char_p
        L.fmo_channel_name.argtypes = [C.c_void_p]
        L.fmo_design_lanczos.restype = C.c_uint
        L.fmo_design_lanczos.argtypes = [C.c_uint, C.c_double, C.c_void_p, C.c_uint]
        L.fmo_design_lp_kaiser.restype = C.c_uint
        L.fmo_design_lp_kaiser.argtypes = [C.c_float] * 5 + [C.c_void_p, C.c_uint]
        L.fmo_design_biquad.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        for name in ("fmo_get_lut", "fmo_get_if_taps", "fmo_get_audio_taps",
                     "fmo_get_rds_lpf_taps", "fmo_get_rds_mf_taps"):
            f = getattr(L, name)
            f.restype = C.c_uint
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.fmo_get_rds_hb_lengths.restype = C.c_uint
        L.fmo_get_rds_hb_lengths.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.fmo_get_constants.restype = C.c_uint
        L.fmo_get_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.fmo_atan2f.restype = C.c_float
        L.fmo_atan2f.argtypes = [C.c_float, C.c_float]
        L.fmo_rds_arctan2.restype = C.c_float
        L.fmo_rds_arctan2.argtypes = [C.c_float, C.c_float]
        L.fmo_convert_u8.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
        L.fmo_receiver_open.restype = C.c_void_p
        L.fmo_receiver_open.argtypes = [C.POINTER(FmoParams), C.c_double, C.c_char_p]
        L.fmo_receiver_close.argtypes = [C.c_void_p]
        L.fmo_receiver_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.fmo_receiver_write_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.fmo_receiver_end.argtypes = [C.c_void_p]
        L.fmo_receiver_queued_samples.restype = C.c_uint64
        L.fmo_receiver_queued_samples.argtypes = [C.c_void_p]
        L.fmo_receiver_set_stream_change.argtypes = [C.c_void_p]
        L.fmo_receiver_demux_read.argtypes = [C.c_void_p, C.POINTER(FmoPacket)]
        L.fmo_receiver_signal_status.argtypes = [C.c_void_p, C.POINTER(C.c_float),
                                                 C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.fmo_receiver_pvr_signal_status.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_char_p,
                                                     C.c_uint, C.c_char_p, C.c_uint,
                                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.fmo_receiver_audio_level.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 3
        L.fmo_sincos_x87.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.fmo_bench_threads.restype = C.c_double
        L.fmo_bench_threads.argtypes = [C.POINTER(FmoParams), C.c_uint, C.c_double, C.c_void_p,
                                        C.c_uint, C.c_uint, C.POINTER(C.c_ulonglong),
                                        C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


CONST_NAMES = [
    "tuning_shift", "demod_gain", "de_alpha", "pll_alpha", "pll_beta", "nco_hl", "nco_ll",
    "pilot_minfreq", "pilot_maxfreq", "pilot_b0", "pilot_a1", "pilot_a2", "pilot_lf_b0",
    "pilot_lf_b1", "pilot_freq", "pilot_lock_delay", "resamp_order", "resamp_step",
    "rds_rate", "rds_nco_inc", "rds_osc_cos", "rds_osc_sin", "rds_pll_alpha", "rds_pll_beta",
    "rds_nco_hl", "rds_nco_ll", "fs_bb", "rds_mf_len",
]


def bench_threads(params, threads, seconds, blocks):
    """fmo_bench_threads: `threads` decoders on native POSIX threads (no Python in the timed loops),
    each replaying blocks [nblocks, 2*samples] float32 for `seconds`.  Returns (IQ samples per
    second over all threads, ProcessStream calls made, longest thread time in s)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.float32)
    nblocks, samples = blocks.shape[0], blocks.shape[1] // 2
    calls, worst = C.c_ulonglong(), C.c_double()
    rate = lib().fmo_bench_threads(C.byref(params), threads, float(seconds), blocks.ctypes.data,
                                   nblocks, samples, C.byref(calls), C.byref(worst))
    if rate <= 0.0:
        raise RuntimeError("fmo_bench_threads failed")
    return rate, int(calls.value), float(worst.value)


def convert_u8(buf):
    """RTL-SDR bytes -> interleaved float32 IQ exactly like cRtlSdrSource::ReadAsyncCB."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    out = np.empty(buf.size, dtype=np.float32)
    lib().fmo_convert_u8(buf.ctypes.data, buf.size // 2, out.ctypes.data)
    return out


class OracleDecoder:
    """Mirror of cFmDecoder's surface on top of the C oracle."""

    def __init__(self, sample_rate_if, tuning_offset, sample_rate_pcm=48000.0,
                 bandwidth_pcm=15000.0, downsample=1, us_version=False, table_size=0,
                 if_filter_order=0, tuning_shift=None):
        p = FmoParams(sample_rate_if, tuning_offset, sample_rate_pcm, bandwidth_pcm, downsample,
                      int(us_version), table_size, if_filter_order,
                      0 if tuning_shift is None else int(tuning_shift),
                      0 if tuning_shift is None else 1)
        self._h = lib().fmo_create(C.byref(p))
        if not self._h:
            raise RuntimeError("fmo_create failed (unsupported configuration)")
        self._audio = np.empty(2 * 65536, dtype=np.float32)

    def close(self):
        if self._h:
            lib().fmo_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def reset(self):
        lib().fmo_reset(self._h)

    def process_stream(self, iq):
        """iq: complex64 or interleaved float32 array; returns interleaved float32 audio."""
        iq = np.ascontiguousarray(iq)
        if iq.dtype == np.complex64:
            iq = iq.view(np.float32)
        assert iq.dtype == np.float32
        n = iq.size // 2
        k = lib().fmo_process_stream(self._h, iq.ctypes.data, n, self._audio.ctypes.data)
        return self._audio[:k].copy()

    def process_stream_u8(self, buf):
        """ReadAsyncCB conversion (RTL_SDR_Source.cpp:207-211) followed by ProcessStream."""
        return self.process_stream(convert_u8(buf))

    def status(self):
        st = FmoStatus()
        lib().fmo_get_status(self._h, C.byref(st))
        return st

    def taps(self):
        t = FmoTaps()
        lib().fmo_get_taps(self._h, C.byref(t))

        def arr(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.float32)

        return {
            "demod": arr(t.demod, 2 * t.n_demod).view(np.complex64),
            "baseband": arr(t.baseband, t.n_demod),
            "pilot38": arr(t.pilot38, t.n_demod),
            "mono_rs": arr(t.mono_rs, t.n_audio),
            "stereo_rs": arr(t.stereo_rs, t.n_audio),
            "rds_lpf": arr(t.rds_lpf, 2 * t.n_rds).view(np.complex64),
            "rds_pll": arr(t.rds_pll, t.n_rds),
            "rds_mf": arr(t.rds_mf, t.n_rds),
            "rds_sync": arr(t.rds_sync, t.n_rds),
        }

    def rds_groups(self):
        out = []
        n = lib().fmo_rds_group_count(self._h)
        blk = (C.c_uint16 * 4)()
        ci = C.c_uint()
        for i in range(n):
            lib().fmo_rds_group_get(self._h, i, blk, C.byref(ci))
            out.append((ci.value, tuple(int(x) for x in blk)))
        return out

    def uecp_frames(self):
        out = []
        n = lib().fmo_uecp_frame_count(self._h)
        buf = (C.c_uint8 * 270)()
        for i in range(n):
            k = lib().fmo_uecp_frame_get(self._h, i, buf, 270)
            out.append(bytes(buf[:k]))
        return out

    def channel_name(self):
        return lib().fmo_channel_name(self._h).decode("latin1")

    def _vec(self, fn, cap=8192, dtype=np.float32):
        buf = np.zeros(cap, dtype=dtype)
        n = getattr(lib(), fn)(self._h, buf.ctypes.data, cap)
        return buf[:n].copy()

    def lut(self):
        return self._vec("fmo_get_lut").view(np.complex64)

    def if_taps(self):
        return self._vec("fmo_get_if_taps")

    def audio_taps(self):
        return self._vec("fmo_get_audio_taps")

    def rds_lpf_taps(self):
        return self._vec("fmo_get_rds_lpf_taps")

    def rds_mf_taps(self):
        return self._vec("fmo_get_rds_mf_taps")

    def rds_hb_lengths(self):
        return [int(x) for x in self._vec("fmo_get_rds_hb_lengths", 16, np.int32)]

    def constants(self):
        v = self._vec("fmo_get_constants", 64, np.float64)
        return dict(zip(CONST_NAMES, v))


class OracleReceiver:
    """The stream members of cRadioReceiver around the oracle decoder (DemuxRead & co)."""

    def __init__(self, sample_rate_if, tuning_offset, downsample, tuner_freq=100.0e6,
                 adapter_name="Generic RTL2832U"):
        p = FmoParams(sample_rate_if, tuning_offset, 48000.0, 15000.0, downsample, 0, 0, 0, 0, 0)
        self._h = lib().fmo_receiver_open(C.byref(p), tuner_freq, adapter_name.encode())
        if not self._h:
            raise RuntimeError("fmo_receiver_open failed")

    def close(self):
        if self._h:
            lib().fmo_receiver_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def write_iq(self, iq):
        iq = np.ascontiguousarray(iq)
        if iq.dtype == np.complex64:
            iq = iq.view(np.float32)
        lib().fmo_receiver_write(self._h, iq.ctypes.data, iq.size // 2)

    def write_u8(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        lib().fmo_receiver_write_u8(self._h, buf.ctypes.data, buf.size // 2)

    def end(self):
        lib().fmo_receiver_end(self._h)

    def queued_samples(self):
        return int(lib().fmo_receiver_queued_samples(self._h))

    def set_stream_change(self):
        lib().fmo_receiver_set_stream_change(self._h)

    def demux_read(self):
        """(stream_id, pts, duration, payload bytes) or None; raises if the reference would block."""
        pkt = FmoPacket()
        rc = lib().fmo_receiver_demux_read(self._h, C.byref(pkt))
        if rc < 0:
            raise RuntimeError("queue empty and end not marked: the reference blocks here")
        if rc == 0:
            return None
        data = bytes(C.string_at(pkt.data, pkt.size)) if pkt.size else b""
        return (pkt.stream_id, pkt.pts, pkt.duration, data)

    def signal_status(self):
        a, b, s = C.c_float(), C.c_float(), C.c_int()
        if not lib().fmo_receiver_signal_status(self._h, C.byref(a), C.byref(b), C.byref(s)):
            return None
        return (a.value, b.value, bool(s.value))

    def pvr_signal_status(self):
        name, status, prov = (C.create_string_buffer(128), C.create_string_buffer(256),
                              C.create_string_buffer(64))
        sig, snr = C.c_int(), C.c_int()
        if not lib().fmo_receiver_pvr_signal_status(self._h, name, 128, status, 256, prov, 64,
                                                    C.byref(sig), C.byref(snr)):
            return None
        return {"adapter_name": name.value.decode(), "adapter_status": status.value.decode(),
                "provider_name": prov.value.decode("latin-1"), "signal": sig.value, "snr": snr.value}

    def audio_level(self):
        m, r, lv = C.c_float(), C.c_float(), C.c_float()
        lib().fmo_receiver_audio_level(self._h, C.byref(m), C.byref(r), C.byref(lv))
        return (m.value, r.value, lv.value)


def design_lanczos(order, cutoff):
    buf = np.zeros(order + 2, np.float32)
    lib().fmo_design_lanczos(order, cutoff, buf.ctypes.data, buf.size)
    return buf


def design_lp_kaiser(scale, astop, fpass, fstop, fs):
    buf = np.zeros(150, np.float32)
    n = lib().fmo_design_lp_kaiser(scale, astop, fpass, fstop, fs, buf.ctypes.data, buf.size)
    return buf[:n].copy()


def design_biquad(ftype, f0, q, fs):
    buf = np.zeros(5, np.float32)
    lib().fmo_design_biquad(ftype, f0, q, fs, buf.ctypes.data)
    return buf


def design_tuner_lut(table_size, freq_shift):
    buf = np.zeros(2 * table_size, np.float32)
    lib().fmo_design_tuner_lut.argtypes = [C.c_uint, C.c_int, C.c_void_p, C.c_uint]
    lib().fmo_design_tuner_lut(table_size, freq_shift, buf.ctypes.data, buf.size)
    return buf

"""pvr.rtl.radiofm_amd -- MI355X-native FM broadcast decoder (hot path of pvr.rtl.radiofm).

The product is the C-ABI shared library ``libfmd_hip.so`` (include/fmd.h) built from csrc/.
This module is only the ctypes plumbing the tests and bench.py use to reach it; the directory
name contains dots, so load it through ``__graft_entry__.load_package()`` (importlib).

There is no CPU fallback here: if the HIP library is missing or no GPU is usable, the calls
raise.  Nothing in this package imports or links oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfmd_hip.so")
_LIB = None

FMD_MAX_BLOCK = 65536
FMD_MIN_BLOCK = 8192
FMD_WARN_RDS_LOST = 1

TAPS = {"demod": 0, "baseband": 1, "pilot38": 2, "mono_rs": 3, "stereo_rs": 4, "rds_lpf": 5,
        "rds_pll": 6, "rds_mf": 7, "rds_sync": 8}
DESIGN = {"if_taps": 0, "rs_taps": 1, "audio_lpf": 2, "rds_lpf": 3, "rds_mf": 4, "scalars": 5,
          "lut0": 6}
SCALAR_NAMES = [
    "tuning_shift", "demod_gain", "de_alpha", "pll_alpha", "pll_beta", "nco_hl", "nco_ll",
    "pilot_minfreq", "pilot_maxfreq", "pilot_b0", "pilot_a1", "pilot_a2", "pilot_lf_b0",
    "pilot_lf_b1", "pilot_freq", "pilot_lock_delay", "resamp_order", "resamp_step",
    "rds_rate", "rds_nco_inc", "rds_osc_cos", "rds_osc_sin", "rds_pll_alpha", "rds_pll_beta",
    "rds_nco_hl", "rds_nco_ll", "fs_bb", "rds_mf_len",
    "notch_b0", "notch_b1", "notch_b2", "notch_a1", "notch_a2",
    "bitsync_b0", "bitsync_b1", "bitsync_b2", "bitsync_a1", "bitsync_a2",
]


class FmdParams(C.Structure):
    _fields_ = [
        ("sample_rate_if", C.c_double),
        ("tuning_offset", C.c_double),
        ("sample_rate_pcm", C.c_double),
        ("bandwidth_pcm", C.c_double),
        ("downsample", C.c_uint),
        ("us_version", C.c_int),
        ("table_size", C.c_uint),
        ("if_filter_order", C.c_uint),
        ("fir_reduction", C.c_int),
    ]


UECP_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint, C.POINTER(C.c_uint8), C.c_uint)
NAME_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint, C.c_char_p)
ACTIVE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint)


class FmdCallbacks(C.Structure):
    _fields_ = [("add_uecp_frame", UECP_CB), ("set_channel_name", NAME_CB),
                ("is_setting_active", ACTIVE_CB)]


class FmdStatus(C.Structure):
    _fields_ = [
        ("stereo_detected", C.c_int),
        ("tuning_offset", C.c_float),
        ("interface_level", C.c_float),
        ("baseband_level", C.c_float),
        ("pilot_level", C.c_float),
        ("rds_state", C.c_int),
    ]


class FmdRdsGroup(C.Structure):
    _fields_ = [("channel", C.c_uint32), ("call_index", C.c_uint32), ("blocks", C.c_uint16 * 4)]


RDS_GROUP_DTYPE = np.dtype([("channel", "<u4"), ("call_index", "<u4"), ("blocks", "<u2", (4,))])
assert RDS_GROUP_DTYPE.itemsize == C.sizeof(FmdRdsGroup)

class FmdAudioLevel(C.Structure):
    _fields_ = [("mean", C.c_float), ("rms", C.c_float), ("level", C.c_float)]


class FmdDemuxPacket(C.Structure):
    _fields_ = [("stream_id", C.c_int), ("size", C.c_int), ("pts", C.c_double),
                ("duration", C.c_double), ("data", C.POINTER(C.c_uint8))]


class FmdPvrSignalStatus(C.Structure):
    _fields_ = [("adapter_name", C.c_char * 128), ("adapter_status", C.c_char * 256),
                ("provider_name", C.c_char * 64), ("signal", C.c_int), ("snr", C.c_int)]


STREAM_AUDIO, STREAM_RDS, STREAM_CHANGE, STREAM_TIME_BASE = 1, 2, -11, 1000000

EXPORTS = [
    "fmd_debug_math", "fmd_batch_debug_serial_probe", "fmd_design_lanczos", "fmd_design_lp_kaiser", "fmd_design_biquad", "fmd_design_tuner_lut",
    "fmd_batch_get_audio_level", "fmd_receiver_open", "fmd_receiver_close", "fmd_receiver_write_iq",
    "fmd_receiver_write_u8", "fmd_receiver_end", "fmd_receiver_queued_samples",
    "fmd_receiver_set_stream_change", "fmd_receiver_demux_read", "fmd_receiver_signal_status",
    "fmd_receiver_pvr_signal_status", "fmd_receiver_decoder",
    "fmd_create", "fmd_destroy", "fmd_reset", "fmd_process_stream", "fmd_process_stream_u8",
    "fmd_get_status", "fmd_batch_process_device_u8", "fmd_batch_process_host_u8",
    "fmd_batch_create", "fmd_batch_destroy", "fmd_batch_reset", "fmd_batch_channels",
    "fmd_batch_min_samples",
    "fmd_batch_max_audio_floats", "fmd_batch_process_device", "fmd_batch_process_host",
    "fmd_batch_collect_rds", "fmd_batch_collect_rds_lagged", "fmd_batch_export_rds_device",
    "fmd_batch_set_concurrency", "fmd_batch_set_channels_per_capture", "fmd_batch_streams_sharing_queue",
    "fmd_batch_wait", "fmd_batch_wait_lagged", "fmd_batch_get_status", "fmd_batch_get_tap", "fmd_batch_get_design",
    "fmd_batch_set_debug_taps", "fmd_batch_set_profiling", "fmd_batch_get_stage_ms", "fmd_stage_name", "fmd_last_error",
    "fmd_version", "fmd_group_decoder_create", "fmd_group_decoder_destroy",
    "fmd_group_decoder_reset", "fmd_group_decoder_push", "fmd_uecp_stuff_frame",
    "fmd_batch_take_rds_lost", "fmd_batch_status_call_index",
    "fmd_batch_debug_set_spin_limit", "fmd_batch_debug_timeline", "fmd_batch_debug_set",
    "fmd_batch_debug_stream_conflicts",
    "fmd_batch_debug_host_ms", "fmd_decoder_batch",
]


def build():
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libfmd_hip.so is not built (run __graft_entry__.build()); "
                               "this package has no CPU fallback")
        # torch bundles its own libamdhip64 (same soname as /opt/rocm's).  Importing torch first
        # makes the loader bind this library to that already-loaded runtime, so device pointers,
        # streams and RCCL from torch and the kernels here share ONE HIP runtime per process.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        vp, u, i = C.c_void_p, C.c_uint, C.c_int
        L.fmd_last_error.restype = C.c_char_p
        L.fmd_version.restype = C.c_char_p
        L.fmd_stage_name.restype = C.c_char_p
        L.fmd_stage_name.argtypes = [u]
        L.fmd_create.argtypes = [C.POINTER(FmdParams), C.POINTER(FmdCallbacks), vp, C.POINTER(vp)]
        L.fmd_destroy.argtypes = [vp]
        L.fmd_reset.argtypes = [vp]
        L.fmd_process_stream.argtypes = [vp, vp, u, vp]
        L.fmd_process_stream_u8.argtypes = [vp, vp, u, vp]
        L.fmd_get_status.argtypes = [vp, C.POINTER(FmdStatus)]
        L.fmd_batch_create.argtypes = [C.POINTER(FmdParams), u, vp, i, C.POINTER(FmdCallbacks), vp,
                                       C.POINTER(vp)]
        L.fmd_batch_destroy.argtypes = [vp]
        L.fmd_batch_reset.argtypes = [vp]
        L.fmd_batch_channels.restype = u
        L.fmd_batch_channels.argtypes = [vp]
        L.fmd_batch_min_samples.restype = u
        L.fmd_batch_min_samples.argtypes = [vp]
        L.fmd_batch_max_audio_floats.restype = u
        L.fmd_batch_max_audio_floats.argtypes = [vp, u]
        L.fmd_batch_process_device.argtypes = [vp, vp, C.c_size_t, u, vp, C.c_size_t,
                                               C.POINTER(u), vp]
        L.fmd_batch_process_host.argtypes = [vp, vp, C.c_size_t, u, vp, C.c_size_t, C.POINTER(u)]
        L.fmd_batch_process_device_u8.argtypes = L.fmd_batch_process_device.argtypes
        L.fmd_batch_process_host_u8.argtypes = L.fmd_batch_process_host.argtypes
        L.fmd_batch_collect_rds.argtypes = [vp, vp, u, i, vp]
        L.fmd_batch_collect_rds_lagged.argtypes = [vp, vp, u, i, i, vp]
        L.fmd_batch_export_rds_device.argtypes = [vp, vp, u, u, i, vp]
        L.fmd_batch_set_concurrency.argtypes = [vp, i]
        L.fmd_batch_set_channels_per_capture.argtypes = [vp, u]
        L.fmd_batch_streams_sharing_queue.argtypes = [vp]
        L.fmd_batch_wait.argtypes = [vp, vp]
        L.fmd_batch_wait_lagged.argtypes = [vp, i, vp]
        L.fmd_batch_get_status.argtypes = [vp, u, C.POINTER(FmdStatus)]
        L.fmd_batch_get_tap.argtypes = [vp, i, u, vp, u]
        L.fmd_batch_get_audio_level.argtypes = [vp, u, C.POINTER(FmdAudioLevel)]
        L.fmd_debug_math.argtypes = [i, u, vp, vp, vp, vp]
        L.fmd_batch_debug_serial_probe.argtypes = [vp, vp, u]
        L.fmd_batch_debug_stream_conflicts.argtypes = [vp, vp]
        L.fmd_design_lanczos.argtypes = [u, C.c_double, vp, u]
        L.fmd_design_lp_kaiser.argtypes = [C.c_float] * 5 + [vp, u]
        L.fmd_design_biquad.argtypes = [i, C.c_float, C.c_float, C.c_float, vp]
        L.fmd_design_tuner_lut.argtypes = [u, i, vp, u]
        L.fmd_receiver_open.argtypes = [C.POINTER(FmdParams), C.c_double, C.c_char_p, C.POINTER(vp)]
        L.fmd_receiver_close.argtypes = [vp]
        L.fmd_receiver_write_iq.argtypes = [vp, vp, u]
        L.fmd_receiver_write_u8.argtypes = [vp, vp, u]
        L.fmd_receiver_end.argtypes = [vp]
        L.fmd_receiver_queued_samples.restype = C.c_size_t
        L.fmd_receiver_queued_samples.argtypes = [vp]
        L.fmd_receiver_set_stream_change.argtypes = [vp]
        L.fmd_receiver_demux_read.argtypes = [vp, C.POINTER(FmdDemuxPacket)]
        L.fmd_receiver_signal_status.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                 C.POINTER(i)]
        L.fmd_receiver_pvr_signal_status.argtypes = [vp, C.POINTER(FmdPvrSignalStatus)]
        L.fmd_receiver_decoder.restype = vp
        L.fmd_receiver_decoder.argtypes = [vp]
        L.fmd_batch_get_design.argtypes = [vp, i, vp, u]
        L.fmd_batch_set_profiling.argtypes = [vp, i]
        L.fmd_batch_set_debug_taps.argtypes = [vp, i]
        L.fmd_batch_get_stage_ms.argtypes = [vp, vp, u]
        L.fmd_group_decoder_create.restype = vp
        L.fmd_group_decoder_create.argtypes = [C.POINTER(FmdCallbacks), vp, u]
        L.fmd_group_decoder_destroy.argtypes = [vp]
        L.fmd_group_decoder_reset.argtypes = [vp]
        L.fmd_group_decoder_push.argtypes = [vp, vp]
        L.fmd_uecp_stuff_frame.argtypes = [vp, u, vp, u]
        L.fmd_batch_take_rds_lost.argtypes = [vp]
        L.fmd_batch_status_call_index.argtypes = [vp, u, C.POINTER(C.c_uint32)]
        L.fmd_batch_debug_set_spin_limit.argtypes = [vp, u]
        L.fmd_batch_debug_timeline.argtypes = [vp, vp, u]
        L.fmd_batch_debug_set.argtypes = [vp, C.c_char_p, i]
        L.fmd_batch_debug_host_ms.argtypes = [vp, vp]
        L.fmd_decoder_batch.restype = vp
        L.fmd_decoder_batch.argtypes = [vp]
        _LIB = L
    return _LIB


class FmdError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise FmdError("fmd error %d: %s" % (rc, lib().fmd_last_error().decode()))
    return rc


FIR_SEQUENTIAL = 0
FIR_SHUFFLE_PARITY_WAIVED = 0x101  # include/fmd.h: the shuffle-reduced IF FIR, outside the parity contract
FIR_FMA_PARITY_WAIVED = 0x102  # FMD_FIR_FMA_PARITY_WAIVED: fused multiply-add, the reference's tap order


def make_params(sample_rate_if, tuning_offset, sample_rate_pcm=48000.0, bandwidth_pcm=15000.0,
                downsample=1, us_version=False, table_size=0, if_filter_order=0, fir_reduction=0):
    return FmdParams(sample_rate_if, tuning_offset, sample_rate_pcm, bandwidth_pcm, downsample,
                     int(us_version), table_size, if_filter_order, fir_reduction)


class _CallbackSink:
    """Python stand-in for the three cRadioReceiver callbacks; records what arrives."""

    def __init__(self):
        self.frames = {}
        self.names = {}
        self.setting_active = False

        def on_frame(_user, ch, data, n):
            self.frames.setdefault(ch, []).append(bytes(data[:n]))
            return 1

        def on_name(_user, ch, name):
            self.names[ch] = name[:8].decode("latin1")
            return 1

        def on_active(_user, ch):
            return 1 if self.setting_active else 0

        self.struct = FmdCallbacks(UECP_CB(on_frame), NAME_CB(on_name), ACTIVE_CB(on_active))


class Batch:
    """C channels of the FM decoder on one GPU (fmd_batch_*)."""

    def __init__(self, params, n_channels, tuning_shifts=None, device=0, record_callbacks=True):
        self.sink = _CallbackSink() if record_callbacks else None
        h = C.c_void_p()
        shifts = None
        if tuning_shifts is not None:
            shifts = np.ascontiguousarray(tuning_shifts, dtype=np.int32)
            assert shifts.size == n_channels
        _check(lib().fmd_batch_create(
            C.byref(params), n_channels, shifts.ctypes.data if shifts is not None else None, device,
            C.byref(self.sink.struct) if self.sink else None, None, C.byref(h)))
        self._h = h
        self.n_channels = n_channels

    def close(self):
        if getattr(self, "_h", None):
            lib().fmd_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def reset(self):
        _check(lib().fmd_batch_reset(self._h))

    def min_samples(self):
        """Smallest call size this batch's geometry accepts (fmd_batch_min_samples)."""
        return lib().fmd_batch_min_samples(self._h)

    def max_audio_floats(self, samples):
        return lib().fmd_batch_max_audio_floats(self._h, samples)

    def process_host(self, iq, shared=False):
        """iq: [C, N] complex64 (or [N] when shared).  Returns [C, n_floats] float32 audio."""
        iq = np.ascontiguousarray(iq)
        if iq.dtype != np.complex64:
            iq = iq.astype(np.float32).view(np.complex64)
        if shared:
            n = iq.size
            stride = 0
        else:
            iq = iq.reshape(self.n_channels // getattr(self, "channels_per_capture", 1), -1)
            n = iq.shape[1]
            stride = n
        a_stride = self.max_audio_floats(n)
        audio = np.zeros((self.n_channels, a_stride), dtype=np.float32)
        nf = C.c_uint()
        _check(lib().fmd_batch_process_host(self._h, iq.ctypes.data, stride, n, audio.ctypes.data,
                                            a_stride, C.byref(nf)))
        return audio[:, :nf.value]

    def process_host_u8(self, iq_u8, shared=False):
        """iq_u8: [C, 2N] uint8 RTL-SDR byte pairs (or [2N] when shared).  Same result as
        process_host on the converted block (RTL_SDR_Source.cpp:207-211)."""
        iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8)
        if shared:
            n = iq_u8.size // 2
            stride = 0
        else:
            iq_u8 = iq_u8.reshape(self.n_channels // getattr(self, "channels_per_capture", 1), -1)
            n = iq_u8.shape[1] // 2
            stride = n
        a_stride = self.max_audio_floats(n)
        audio = np.zeros((self.n_channels, a_stride), dtype=np.float32)
        nf = C.c_uint()
        _check(lib().fmd_batch_process_host_u8(self._h, iq_u8.ctypes.data, stride, n,
                                               audio.ctypes.data, a_stride, C.byref(nf)))
        return audio[:, :nf.value]

    def process_device(self, d_iq_ptr, iq_stride, samples, d_audio_ptr, audio_stride, stream=None,
                       u8=False):
        """iq_stride in IQ samples; u8=True: d_iq_ptr holds RTL-SDR byte pairs."""
        nf = C.c_uint()
        fn = lib().fmd_batch_process_device_u8 if u8 else lib().fmd_batch_process_device
        _check(fn(self._h, d_iq_ptr, iq_stride, samples, d_audio_ptr, audio_stride, C.byref(nf),
                  stream))
        return nf.value

    def collect_rds_array(self, cap=65536, run_group_decoder=False, stream=None, lag=0):
        """Queued RDS groups as a numpy structured array (channel, call_index, blocks[4])."""
        if getattr(self, "_rds_buf", None) is None or self._rds_buf.size < cap:
            self._rds_buf = np.zeros(cap, dtype=RDS_GROUP_DTYPE)
        n = _check(lib().fmd_batch_collect_rds_lagged(self._h, self._rds_buf.ctypes.data, cap,
                                                      int(run_group_decoder), lag, stream))
        return self._rds_buf[:n].copy()

    def collect_rds(self, cap=65536, run_group_decoder=False, stream=None, lag=0):
        """Queued RDS groups as a list of (channel, call_index, (b0, b1, b2, b3))."""
        a = self.collect_rds_array(cap, run_group_decoder, stream, lag)
        return [(int(c), int(k), tuple(int(x) for x in b))
                for c, k, b in zip(a["channel"], a["call_index"], a["blocks"])]

    def export_rds_device(self, d_records_ptr, cap, channel_offset=0, stream=None, lag=0):
        """Queued RDS groups of calls at least `lag` old -> [cap, 4] int32 rows in device memory
        (fmd_batch_export_rds_device), asynchronously on `stream`.  True: groups were lost."""
        return _check(lib().fmd_batch_export_rds_device(self._h, d_records_ptr, cap, channel_offset, lag,
                                                        stream)) == FMD_WARN_RDS_LOST

    def streams_sharing_queue(self):
        """Internal streams that share a hardware queue with another stream of the process (0: none)."""
        return lib().fmd_batch_streams_sharing_queue(self._h)

    def set_channels_per_capture(self, k):
        """k consecutive channels tune the same capture; the process calls then take one input row per capture."""
        _check(lib().fmd_batch_set_channels_per_capture(self._h, int(k)))
        self.channels_per_capture = max(1, int(k))

    def set_concurrency(self, mode):
        _check(lib().fmd_batch_set_concurrency(self._h, int(mode)))

    def wait(self, stream=None, lag=0):
        """Returns True when RDS groups were lost since the last report (FMD_WARN_RDS_LOST)."""
        return _check(lib().fmd_batch_wait_lagged(self._h, lag, stream)) == FMD_WARN_RDS_LOST

    def take_rds_lost(self):
        return bool(_check(lib().fmd_batch_take_rds_lost(self._h)))

    def status_call_index(self, channel=0):
        """Index of the call whose status the getters return right now (0 = none yet)."""
        ci = C.c_uint32()
        _check(lib().fmd_batch_status_call_index(self._h, channel, C.byref(ci)))
        return ci.value

    def debug_timeline(self, cap=512):
        """[calls, 10] ms since the first profiled call's FIR start: FIR, serial stage, audio tail,
        half-band chain, resampler (start, end each); empty unless calls overlap at profiling level 1."""
        buf = np.full((cap, 10), -1.0, dtype=np.float32)
        n = _check(lib().fmd_batch_debug_timeline(self._h, buf.ctypes.data, cap))
        return buf[:n].copy()

    def debug_stream_conflicts(self, stream=None):
        """Bit mask of internal streams that share a hardware queue with `stream` (0: none); drains the device."""
        return _check(lib().fmd_batch_debug_stream_conflicts(self._h, stream))

    def debug_set_spin_limit(self, limit):
        _check(lib().fmd_batch_debug_set_spin_limit(self._h, limit))

    def debug_set(self, key, value):
        """Development switch of this batch by name (fmd_batch_debug_set)."""
        _check(lib().fmd_batch_debug_set(self._h, key.encode(), int(value)))

    def debug_host_ms(self):
        """(calls, {copy_in, submit, wait_copy_out, rds_callbacks}) mean ms of the host-buffer calls since
        the last query (fmd_batch_debug_host_ms)."""
        out = (C.c_float * 4)()
        n = _check(lib().fmd_batch_debug_host_ms(self._h, out))
        return n, dict(zip(("copy_in", "submit", "wait_copy_out", "rds_callbacks"), (float(v) for v in out)))

    def status(self, channel=0):
        st = FmdStatus()
        _check(lib().fmd_batch_get_status(self._h, channel, C.byref(st)))
        return st

    def audio_level(self, channel=0):
        """(audio_mean, audio_rms, m_AudioLevel) of cRadioReceiver::DemuxRead for one channel."""
        a = FmdAudioLevel()
        _check(lib().fmd_batch_get_audio_level(self._h, channel, C.byref(a)))
        return (a.mean, a.rms, a.level)

    def debug_serial_probe(self):
        """FMD_SERIAL_PROBE=1: (start, end, cycles) int64 per workgroup of the serial stage's last 8
        launches, shape (8, slots, 3), launch = call index mod 8; unused slots are zero."""
        buf = np.zeros((8 * 4096, 3), dtype=np.int64)
        n = _check(lib().fmd_batch_debug_serial_probe(self._h, buf.ctypes.data, buf.shape[0]))
        n -= n % 8
        return buf[:n].reshape(8, -1, 3).copy() if n else buf[:0].reshape(8, 0, 3)

    def tap(self, name, channel=0):
        cap = 2 * 65536
        buf = np.zeros(cap, dtype=np.float32)
        n = _check(lib().fmd_batch_get_tap(self._h, TAPS[name], channel, buf.ctypes.data, cap))
        if name in ("demod", "rds_lpf"):
            return buf[:2 * n].view(np.complex64).copy()
        return buf[:n].copy()

    def enable_taps(self, on=True):
        _check(lib().fmd_batch_set_debug_taps(self._h, int(on)))

    def design(self, name):
        buf = np.zeros(16384, dtype=np.float32)
        n = _check(lib().fmd_batch_get_design(self._h, DESIGN[name], buf.ctypes.data, buf.size))
        return buf[:n].copy()

    def scalars(self):
        return dict(zip(SCALAR_NAMES, self.design("scalars")))

    def set_profiling(self, level=2):
        _check(lib().fmd_batch_set_profiling(self._h, int(level)))

    def stage_ms(self):
        """(average ms per stage, number of calls averaged); stages not covered read -1."""
        buf = np.full(32, -1.0, dtype=np.float32)
        calls = _check(lib().fmd_batch_get_stage_ms(self._h, buf.ctypes.data, buf.size))
        out = {}
        i = 0
        while True:
            name = lib().fmd_stage_name(i).decode()
            if not name:
                break
            out[name] = float(buf[i])
            i += 1
        return out, calls


class _BatchView(Batch):
    """A batch somebody else owns (FmDecoder.batch_view)."""

    def __init__(self, handle, sink):
        self._h = handle
        self.sink = sink
        self.n_channels = 1

    def close(self):
        self._h = None

    def __del__(self):
        pass


class FmDecoder:
    """The reference's cFmDecoder surface (FmDecode.h:110-165) on the GPU library."""

    def __init__(self, sample_rate_if, tuning_offset, sample_rate_pcm, bandwidth_pcm=15000.0,
                 downsample=1, USver=False):
        self.sink = _CallbackSink()
        p = make_params(sample_rate_if, tuning_offset, sample_rate_pcm, bandwidth_pcm, downsample,
                        USver)
        h = C.c_void_p()
        _check(lib().fmd_create(C.byref(p), C.byref(self.sink.struct), None, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().fmd_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def Reset(self):
        _check(lib().fmd_reset(self._h))

    def batch_view(self):
        """The decoder's one-channel batch as a Batch object that does not own it: for the profiling and
        development calls (set_profiling, stage_ms, debug_host_ms, debug_set), not for processing."""
        return _BatchView(C.c_void_p(lib().fmd_decoder_batch(self._h)), self.sink)

    def ProcessStream(self, samples_in):
        iq = np.ascontiguousarray(samples_in)
        if iq.dtype != np.complex64:
            iq = iq.astype(np.float32).view(np.complex64)
        audio = np.empty(2 * iq.size, dtype=np.float32)  # RadioReceiver.cpp:519-520 sizing
        n = _check(lib().fmd_process_stream(self._h, iq.ctypes.data, iq.size, audio.ctypes.data))
        return audio[:n]

    def ProcessStreamU8(self, buf):
        """ReadAsyncCB + ProcessStream (RTL_SDR_Source.cpp:196-213): buf = I,Q byte pairs."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        n_iq = buf.size // 2
        audio = np.empty(2 * n_iq, dtype=np.float32)
        n = _check(lib().fmd_process_stream_u8(self._h, buf.ctypes.data, n_iq, audio.ctypes.data))
        return audio[:n]

    def _status(self):
        st = FmdStatus()
        _check(lib().fmd_get_status(self._h, C.byref(st)))
        return st

    def StereoDetected(self):
        return bool(self._status().stereo_detected)

    def GetTuningOffset(self):
        return self._status().tuning_offset

    def GetInterfaceLevel(self):
        return self._status().interface_level

    def GetBasebandLevel(self):
        return self._status().baseband_level

    def GetPilotLevel(self):
        return self._status().pilot_level


class GroupDecoder:
    """Host-only UECP group decoder (fmd_group_decoder_*)."""

    def __init__(self):
        self.sink = _CallbackSink()
        self._h = lib().fmd_group_decoder_create(C.byref(self.sink.struct), None, 0)

    def push(self, blocks):
        b = (C.c_uint16 * 4)(*blocks)
        lib().fmd_group_decoder_push(self._h, b)

    def reset(self):
        lib().fmd_group_decoder_reset(self._h)

    @property
    def frames(self):
        return self.sink.frames.get(0, [])

    @property
    def name(self):
        return self.sink.names.get(0, "")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fmd_group_decoder_destroy(self._h)
            self._h = None


def stuff_uecp_frame(frame):
    out = (C.c_uint8 * (2 * len(frame) + 2))()
    src = (C.c_uint8 * len(frame))(*frame)
    n = lib().fmd_uecp_stuff_frame(src, len(frame), out, len(out))
    return bytes(out[:n])


class Receiver:
    """The stream members of cRadioReceiver around the GPU decoder: WriteDataBuffer,
    EndDataBuffer, DemuxRead, GetSignalStatus (RadioReceiver.cpp:296-349, 387-582)."""

    def __init__(self, sample_rate_if, tuning_offset, downsample, tuner_freq=100.0e6,
                 adapter_name="Generic RTL2832U"):
        p = make_params(sample_rate_if, tuning_offset, 48000.0, 15000.0, downsample)
        h = C.c_void_p()
        _check(lib().fmd_receiver_open(C.byref(p), tuner_freq, adapter_name.encode(), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().fmd_receiver_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def write_iq(self, iq):
        iq = np.ascontiguousarray(iq)
        if iq.dtype != np.complex64:
            iq = iq.astype(np.float32).view(np.complex64)
        _check(lib().fmd_receiver_write_iq(self._h, iq.ctypes.data, iq.size))

    def write_u8(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        _check(lib().fmd_receiver_write_u8(self._h, buf.ctypes.data, buf.size // 2))

    def end(self):
        lib().fmd_receiver_end(self._h)

    def queued_samples(self):
        return int(lib().fmd_receiver_queued_samples(self._h))

    def set_stream_change(self):
        lib().fmd_receiver_set_stream_change(self._h)

    def demux_read(self):
        """(stream_id, pts, duration, payload bytes) or None (the reference's nullptr)."""
        pkt = FmdDemuxPacket()
        rc = _check(lib().fmd_receiver_demux_read(self._h, C.byref(pkt)))
        if rc == 0:
            return None
        data = bytes(C.string_at(pkt.data, pkt.size)) if pkt.size else b""
        return (pkt.stream_id, pkt.pts, pkt.duration, data)

    def signal_status(self):
        a, b, s = C.c_float(), C.c_float(), C.c_int()
        rc = _check(lib().fmd_receiver_signal_status(self._h, C.byref(a), C.byref(b), C.byref(s)))
        return (a.value, b.value, bool(s.value)) if rc else None

    def pvr_signal_status(self):
        st = FmdPvrSignalStatus()
        rc = _check(lib().fmd_receiver_pvr_signal_status(self._h, C.byref(st)))
        if not rc:
            return None
        return {"adapter_name": st.adapter_name.decode(), "adapter_status": st.adapter_status.decode(),
                "provider_name": st.provider_name.decode("latin-1"), "signal": st.signal,
                "snr": st.snr}


# ---- filter design on the host (no GPU needed), same constructors as the reference ----------
def design_lanczos(order, cutoff):
    """cDownsampleFilter's Lanczos table (DownConvert.cpp:18-56, :78): order + 2 floats."""
    buf = np.zeros(order + 2, np.float32)
    _check(lib().fmd_design_lanczos(order, cutoff, buf.ctypes.data, buf.size))
    return buf


def design_lp_kaiser(scale, astop, fpass, fstop, fs):
    """cFirFilter::InitLPFilter (FirFilter.cpp:44-140)."""
    buf = np.zeros(4096, np.float32)
    n = _check(lib().fmd_design_lp_kaiser(scale, astop, fpass, fstop, fs, buf.ctypes.data, buf.size))
    return buf[:n].copy()


def design_biquad(ftype, f0, q, fs):
    """cIirFilter::Init (IirFilter.cpp:11-60): b0 b1 b2 a1 a2; ftype 0 LP, 1 HP, 2 BP, 3 BR."""
    buf = np.zeros(5, np.float32)
    _check(lib().fmd_design_biquad(ftype, f0, q, fs, buf.ctypes.data))
    return buf


def design_tuner_lut(table_size, freq_shift):
    """cFineTuner's table (FmDecode.cpp:45-58), interleaved re, im."""
    buf = np.zeros(2 * table_size, np.float32)
    _check(lib().fmd_design_tuner_lut(table_size, freq_shift, buf.ctypes.data, buf.size))
    return buf


def debug_math(what, a, b=None):
    """Device build of one csrc/fmd_math.h helper on arrays (see fmd_debug_math)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    o0, o1 = np.empty_like(a), np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float32)
        bp = b.ctypes.data
    _check(lib().fmd_debug_math(what, a.size, a.ctypes.data, bp, o0.ctypes.data, o1.ctypes.data))
    return o0, o1

"""ctypes binding of libfmd_gather.so (include/fmd_gather.h): the rank-0 gather of float audio and RDS
records over RCCL, written in C++ (csrc/fmd_gather.hip).  bench.py uses it for the data path when the
backend is RCCL; torch.distributed stays for the rendezvous (it carries the communicator's id)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfmd_gather.so")
ID_BYTES = 128
EXPORTS = ["fmd_gather_last_error", "fmd_gather_unique_id", "fmd_gather_create", "fmd_gather_destroy",
           "fmd_gather_step", "fmd_gather_step_root", "fmd_gather_wait", "fmd_gather_wait_lagged", "fmd_gather_barrier",
           "fmd_gather_ms_per_step", "fmd_gather_info", "fmd_gather_debug_emulate_peers",
           "fmd_gather_debug_emulate_role"]
_LIB = None


class GatherInfo(C.Structure):
    _fields_ = [("ranks_seen", C.c_int), ("rank", C.c_int), ("device", C.c_int), ("world_asked", C.c_int),
                ("steps_issued", C.c_uint64)]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libfmd_gather.so is not built (run __graft_entry__.build())")
        import torch  # noqa: F401  (its HIP runtime and RCCL first: one of each per process)
        L = C.CDLL(LIB_PATH)
        vp, u, i = C.c_void_p, C.c_uint, C.c_int
        L.fmd_gather_last_error.restype = C.c_char_p
        L.fmd_gather_unique_id.argtypes = [vp]
        L.fmd_gather_create.argtypes = [vp, i, i, i, C.c_size_t, u, C.POINTER(vp)]
        L.fmd_gather_destroy.argtypes = [vp]
        L.fmd_gather_step.argtypes = [vp, vp, i, u, vp, vp, vp, vp, vp]
        L.fmd_gather_step_root.argtypes = [vp, i, vp, i, u, vp, vp, vp, vp, vp]
        L.fmd_gather_wait.argtypes = [vp, vp]
        L.fmd_gather_wait_lagged.argtypes = [vp, u, vp]
        L.fmd_gather_barrier.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
        L.fmd_gather_ms_per_step.restype = C.c_float
        L.fmd_gather_ms_per_step.argtypes = [vp]
        L.fmd_gather_info.argtypes = [vp, C.POINTER(GatherInfo)]
        L.fmd_gather_debug_emulate_peers.argtypes = [vp, i, i]
        L.fmd_gather_debug_emulate_role.argtypes = [vp, i]
        _LIB = L
    return _LIB


class GatherError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise GatherError("fmd_gather error %d: %s" % (rc, lib().fmd_gather_last_error().decode()))
    return rc


def unique_id():
    """bytes(128), on rank 0 (ncclGetUniqueId)."""
    buf = (C.c_uint8 * ID_BYTES)()
    _check(lib().fmd_gather_unique_id(buf))
    return bytes(buf)


class Gather:
    def __init__(self, uid, rank, world, device, audio_floats, rds_rows):
        h = C.c_void_p()
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        _check(lib().fmd_gather_create(buf, rank, world, device, audio_floats, rds_rows, C.byref(h)))
        self._h = h
        self.issued = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().fmd_gather_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def step(self, batch, lag, channel_offset, d_audio, d_rds, d_all_audio, d_all_rds, stream, root=0):
        """Returns the step's ticket (for wait_for).  root: the rank that receives this step (fmd_gather_step_root)."""
        _check(lib().fmd_gather_step_root(self._h, int(root), batch._h if batch is not None else None, lag,
                                          channel_offset, d_audio, d_rds, d_all_audio, d_all_rds, stream))
        self.issued += 1
        return self.issued - 1

    def wait_for(self, ticket, stream):
        """Orders `stream` behind the step with this ticket (steps complete in order)."""
        lag = self.issued - 1 - ticket
        _check(lib().fmd_gather_wait_lagged(self._h, max(0, min(lag, 15)), stream))

    def wait(self, stream):
        _check(lib().fmd_gather_wait(self._h, stream))

    def barrier(self, value=0.0):
        m = C.c_double()
        _check(lib().fmd_gather_barrier(self._h, value, C.byref(m)))
        return m.value

    def ms_per_step(self):
        v = lib().fmd_gather_ms_per_step(self._h)
        return None if v < 0 else float(v)

    def emulate_peers(self, peers, workgroups_per_peer=2):
        """Measurement aid (world of one): every step also writes what `peers` more ranks' receives would write."""
        _check(lib().fmd_gather_debug_emulate_peers(self._h, int(peers), int(workgroups_per_peer)))

    def emulate_role(self, every):
        """1: the root of every step; peers + 1: a rank of a rotating root; 0: a sender in every step."""
        _check(lib().fmd_gather_debug_emulate_role(self._h, int(every)))

    def info(self):
        """What the communicator itself reports: ranks_seen (ncclCommCount), rank, device, steps issued."""
        out = GatherInfo()
        _check(lib().fmd_gather_info(self._h, C.byref(out)))
        return {"ranks_seen": out.ranks_seen, "rank": out.rank, "device": out.device,
                "world_asked": out.world_asked, "steps_issued": int(out.steps_issued)}

"""Gather of per-rank decoder outputs to rank 0 (torch.distributed; RCCL on GPUs, gloo in tests).

Channels are sharded across ranks with no exchange during compute; the only
communication is this gather of float audio and RDS group records, once per step.  Records are
fixed-size int32 rows [channel+1, call_index, b0|b1<<16, b2|b3<<16], zero rows = padding, so the
message size is the same on every rank and step.
"""
import numpy as np

RDS_REC_WIDTH = 4


def pack_rds_records(groups, cap, channel_offset=0):
    """groups: structured array (channel, call_index, blocks[4]) -> int32 [cap, 4]."""
    rec = np.zeros((cap, RDS_REC_WIDTH), dtype=np.int32)
    n = int(groups.size)
    if n > cap:  # a fixed-size message cannot carry them: never drop groups silently
        raise ValueError("pack_rds_records: %d groups do not fit into %d records" % (n, cap))
    if n:
        g = groups[:n]
        b = g["blocks"].astype(np.int64)
        rec[:n, 0] = g["channel"].astype(np.int64) + 1 + channel_offset
        rec[:n, 1] = g["call_index"]
        rec[:n, 2] = (b[:, 0] | (b[:, 1] << 16)).astype(np.uint32).view(np.int32)
        rec[:n, 3] = (b[:, 2] | (b[:, 3] << 16)).astype(np.uint32).view(np.int32)
    return rec


def unpack_rds_records(rec):
    """int32 [n, 4] -> list of (global channel, call_index, (b0, b1, b2, b3)), padding dropped."""
    rec = np.asarray(rec)
    out = []
    for ch1, ci, w0, w1 in rec[rec[:, 0] > 0]:
        w0 = int(np.uint32(w0))
        w1 = int(np.uint32(w1))
        out.append((int(ch1) - 1, int(ci), (w0 & 0xFFFF, w0 >> 16, w1 & 0xFFFF, w1 >> 16)))
    return out


def gather_step(audio, rds_rec, gather_audio=None, gather_rds=None, dst=0, async_op=False):
    """Gather this rank's audio [C, stride] and RDS records [cap, 4] tensors to rank `dst`.

    gather_audio / gather_rds: lists of world_size receive tensors on rank dst, None elsewhere.
    Returns the two work handles (async_op) or None."""
    import torch.distributed as dist
    rank = dist.get_rank()
    w1 = dist.gather(audio, gather_audio if rank == dst else None, dst=dst, async_op=async_op)
    w2 = dist.gather(rds_rec, gather_rds if rank == dst else None, dst=dst, async_op=async_op)
    return (w1, w2) if async_op else None


def gather_preflight(device):
    """One tiny gather to rank 0 and a check of what arrived, before any buffer is sized for the
    real thing: raises on every rank if the backend cannot gather (so a launch fails at once and
    with a message instead of inside the timed region)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    x = torch.full((8,), float(rank + 1), dtype=torch.float32, device=device)
    out = [torch.zeros_like(x) for _ in range(world)] if rank == 0 else None
    dist.gather(x, out, dst=0)
    ok = torch.ones(1, dtype=torch.float32, device=device)
    if rank == 0:
        good = all(bool((out[r] == float(r + 1)).all()) for r in range(world))
        ok.fill_(1.0 if good else 0.0)
    dist.broadcast(ok, src=0)
    if float(ok.item()) != 1.0:
        raise RuntimeError("gather pre-flight: rank 0 did not receive every rank's tensor")

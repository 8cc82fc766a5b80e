"""The N > 1 output path (rank-0 gather of float audio + RDS records) on CPU with gloo,
world_size 2: packing, channel offsets, padding and ordering."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from __graft_entry__ import ROOT, load_package


def test_pack_unpack_roundtrip():
    pkg = load_package()
    import importlib
    dg = importlib.import_module(pkg.__name__ + ".dist_gather")
    g = np.zeros(5, dtype=pkg.RDS_GROUP_DTYPE)
    g["channel"] = [0, 3, 3, 7, 8191]
    g["call_index"] = [1, 1, 2, 9, 65536]
    g["blocks"] = [[0xD314, 0x0148, 0xE0CD, 0x5445], [1, 2, 3, 4], [0xFFFF, 0, 0xFFFF, 0],
                   [0x8000, 0x8000, 0x8000, 0x8000], [5, 6, 7, 8]]
    rec = dg.pack_rds_records(g, 8, channel_offset=8192)
    assert rec.shape == (8, 4) and (rec[5:] == 0).all()
    back = dg.unpack_rds_records(rec)
    assert back == [(int(c) + 8192, int(k), tuple(int(x) for x in b))
                    for c, k, b in zip(g["channel"], g["call_index"], g["blocks"])]
    import pytest
    with pytest.raises(ValueError):  # more groups than records: refused, never dropped silently
        dg.pack_rds_records(g, 3)


WORKER = textwrap.dedent("""
    import os, sys, importlib
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from __graft_entry__ import load_package
    pkg = load_package()
    dg = importlib.import_module(pkg.__name__ + ".dist_gather")
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
    rank = dist.get_rank()
    dg.gather_preflight("cpu")
    C, stride, cap = 6, 16, 8
    audio = torch.arange(C * stride, dtype=torch.float32).reshape(C, stride) + 1000.0 * rank
    g = np.zeros(2 + rank, dtype=pkg.RDS_GROUP_DTYPE)
    g["channel"] = np.arange(g.size)
    g["call_index"] = 4
    g["blocks"] = (np.arange(4 * g.size).reshape(-1, 4) + 100 * rank)
    rec = torch.from_numpy(dg.pack_rds_records(g, cap, channel_offset=rank * C))
    ga = [torch.empty_like(audio) for _ in range(2)] if rank == 0 else None
    gr = [torch.empty_like(rec) for _ in range(2)] if rank == 0 else None
    for _ in range(3):  # three steps, like the bench loop
        w = dg.gather_step(audio, rec, ga, gr, dst=0, async_op=True)
        for x in w:
            x.wait()
    if rank == 0:
        assert torch.equal(ga[0], audio)
        assert torch.equal(ga[1], audio + 1000.0)
        got = [dg.unpack_rds_records(r.numpy()) for r in gr]
        assert [c for c, _, _ in got[0]] == [0, 1]
        assert [c for c, _, _ in got[1]] == [6, 7, 8]
        assert got[1][2][2] == (108, 109, 110, 111)
        print("GATHER_OK")
    dist.barrier()
    dist.destroy_process_group()
""")


def test_world_size_2_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        procs.append(subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(rank)),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]

"""The whole-node gather (include/fmd_gather.h, csrc/fmd_gather.hip) with a world of 2, 3 and 8 ranks in
the build container.  No N > 1 hardware run has been possible on this pool, and a world of one issues
neither ncclSend nor ncclRecv -- so the product's gather source is compiled here with g++ and linked against
a TEST DOUBLE of the HIP-runtime and RCCL calls it makes (tests/cpp/gather_double/fake_hip_rccl.cpp: streams
as in-order worker queues, events, grouped send / receive over shared-memory rings between processes) in
place of libamdhip64 / librccl.  tests/cpp/gather_double/world_n.cpp forks the ranks, produces every step's
outputs late on the caller's stream, rotates six buffers over 40 steps (the 16-event ring wraps) and has
rank 0 compare every byte it received with what rank r must have sent.

The double replaces libraries under the product's source for this test only; nothing of it is linked into,
or loaded by, the product.  (Mutations of fmd_gather.hip that drop the ordering behind the caller's stream
or shift a receive offset make this test fail: checked when it was written.)"""
import json
import os
import subprocess

import pytest

from __graft_entry__ import ROOT

HERE = os.path.join(ROOT, "tests", "cpp", "gather_double")


@pytest.fixture(scope="module")
def world_n(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gather_double") / "world_n")
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-x", "c++",
           os.path.join(ROOT, "pvr.rtl.radiofm_amd", "csrc", "fmd_gather.hip"),
           os.path.join(HERE, "fake_hip_rccl.cpp"), os.path.join(HERE, "world_n.cpp"), "-o", exe, "-lpthread", "-lrt"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


@pytest.mark.parametrize("world,steps", [(2, 40), (3, 40), (8, 20)])
def test_gather_step_with_real_peers_against_the_double(world_n, world, steps):
    out = subprocess.run([world_n, str(world), str(steps)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["world"] == world and line["rccl_ranks_seen"] == world
    assert line["rank_step_messages_checked"] == world * steps


def test_rank0_outputs_produced_in_place(world_n):
    """Rank 0's own audio / records written straight into its part of the receive buffers (what bench.py and
    tools/node_bench do): the gather copies nothing for it and the peers' parts still land beside it."""
    env = dict(os.environ, WORLD_N_INPLACE="1")
    out = subprocess.run([world_n, "3", "40"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["rank_step_messages_checked"] == 120


def test_a_failed_receive_reports_and_closes_its_group(world_n):
    env = dict(os.environ, FAKE_RCCL_FAIL_RECV="1")
    out = subprocess.run([world_n, "2", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]


def test_a_stalled_bootstrap_restarts_all_ranks_once(world_n):
    """RCCL's bootstrap stalls about once in 20 launches on this pool (no error: every rank waits inside
    ncclCommInitRank until its watchdog ends it).  The parent of a whole-node run (tools/rank_supervisor.hpp: what
    tools/node_bench uses; bench.py has the same per rank) has not touched HIP: when not every rank reports
    "communicator up" in time it ends the ranks and starts all of them again, once, as fresh children.  Here rank 1's
    first ncclCommInitRank never returns; the second attempt runs to the end with every byte checked."""
    env = dict(os.environ, FAKE_RCCL_STALL_INIT="1", WORLD_N_UP_TIMEOUT="2")
    out = subprocess.run([world_n, "3", "8"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "starting all of them again, once" in out.stderr and "world_n: attempts 2" in out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["world"] == 3 and line["rccl_ranks_seen"] == 3 and line["rank_step_messages_checked"] == 24
    # no stall: one attempt
    out = subprocess.run([world_n, "2", "4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "world_n: attempts 1" in out.stderr


@pytest.mark.parametrize("world,steps", [(3, 40), (8, 24)])
def test_rotating_root_spreads_the_receive_load(world_n, world, steps):
    """fmd_gather_step_root: step i is gathered to rank i % world -- every rank is the receiver of every world-th step
    (with its own part produced in place in ITS slot of its receive buffers) and a sender otherwise; what rank 0 alone
    would take (7 x 88 MB of writes per step at 8 GPUs: 8-10 % of that GPU's throughput, docs/MEASUREMENTS.md) is
    spread evenly.  Every root checks every byte of the steps it received."""
    env = dict(os.environ, WORLD_N_ROTATE="1")
    out = subprocess.run([world_n, str(world), str(steps)], capture_output=True, text=True, timeout=180, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == list(range(world))
    assert all(l["rccl_ranks_seen"] == world for l in lines)
    assert sum(l["rank_step_messages_checked"] for l in lines) == world * steps
    for l in lines:  # a rank is the root of the steps i with i % world == rank
        assert l["rank_step_messages_checked"] == world * len(range(l["rank"], steps, world))

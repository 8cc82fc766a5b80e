"""`python bench.py --gpus N` with N > 1 and no rank variables in the environment starts its own N
ranks (one child process per GPU, before anything touches HIP).  Without a GPU the ranks cannot get
past device selection: what is checked here is the launcher -- it comes back promptly with a nonzero
exit code and names the ranks that failed, instead of hanging in a rendezvous.  The same invocation on a
GPU box is tests/test_gpu_multirank.py::test_bench_starts_its_own_ranks."""
import os
import subprocess
import sys

from __graft_entry__ import ROOT


def test_launcher_reports_failed_ranks_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by the gpu test of the same invocation")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--channels", "64", "--no-cpu-baseline", "--watchdog", "60"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "ranks failed" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]  # no result line


def test_launcher_takes_its_ranks_with_it_when_it_is_terminated():
    """A driver's timeout or a cancelled job sends the launcher SIGTERM: the ranks it started (here: two
    that never finish) must be gone with it, not left holding the GPUs until their own watchdog fires."""
    import re
    import signal
    import time
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FMD_BENCH_TEST_HANG"] = "1"
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                         env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = ""
    t_end = time.time() + 30
    while "started ranks" not in line and time.time() < t_end:
        line = p.stderr.readline()
    pids = [int(x) for x in re.findall(r"\d+", line.split("pids", 1)[1])]
    assert len(pids) == 2
    time.sleep(0.5)
    assert all(os.path.exists("/proc/%d" % q) for q in pids)
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=15)
    assert rc == 128 + signal.SIGTERM
    t_end = time.time() + 5
    def gone(q):
        try:
            return open("/proc/%d/stat" % q).read().split()[2] == "Z"
        except OSError:
            return True
    while not all(gone(q) for q in pids) and time.time() < t_end:
        time.sleep(0.05)
    assert all(gone(q) for q in pids)


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LD_PRELOAD")}
    env.update(extra)
    return env


def test_a_stalled_communicator_restarts_every_rank_once():
    """RCCL's bootstrap stalls about once in 20 launches on this pool.  Every rank of an N > 1 bench.py run is a
    supervisor (no torch, no HIP) over a worker child; a worker whose communicator is not up 60 s (here 2 s) after
    the host-side rendezvous is ended and started again, once, with a rendezvous of its own (_supervise_rank).
    FMD_BENCH_TEST_SUPERVISOR=1 replaces the worker's body by one that never reports "up" on its first attempt."""
    import json
    env = _clean_env(FMD_BENCH_TEST_SUPERVISOR="1", FMD_BENCH_UP_TIMEOUT="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline",
                          "--watchdog", "60"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stderr.count("starting it again, once") == 2  # one line per rank
    d = json.loads(out.stdout.strip().splitlines()[-1])  # rank 0's second worker
    assert d["test_worker"] and d["attempt"] == 1 and d["rank"] == 0


def test_the_restart_under_torch_distributed_run():
    """The driver's N > 1 invocation: torch.distributed.run starts the ranks, each becomes supervisor + worker; the
    restarted workers meet under a rendezvous prefix of their own on the agent's store (or, where rank 0's worker
    hosts the store, on the next port)."""
    import json
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = _clean_env(FMD_BENCH_TEST_SUPERVISOR="1", FMD_BENCH_UP_TIMEOUT="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--watchdog", "60"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1] and all(l["attempt"] == 1 for l in lines)
    assert all(l["restart_count"] == "100" or int(l["master_port"]) == port + 1 for l in lines)

#!/usr/bin/env python3
"""bench.py -- IQ MS/s demodulated by the MI355X FM decoder, whole job, with the FIR-stage roofline.

    python bench.py --gpus N --steps K --warmup W          (any N: with N > 1 and no WORLD_SIZE in the
                                                            environment it starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[3], per-GPU shard): 8192 independent synthetic FM stereo+RDS
channels per GPU at 2.4 MS/s (65 536 over 8 GPUs), one step = one ProcessStream call of 65 536
IQ samples on every channel: tuner mix -> 88-tap decimating FIR -> FM PLL -> pilot PLL / stereo ->
RDS chain -> resamplers -> audio filters -> float stereo audio + RDS groups.  Inputs are generated
on the device and are resident in HBM before the timed region.  With N > 1 every rank owns its
own channels (no exchange during compute) and float audio and RDS groups of every step are gathered
over RCCL -- step i to rank i % N (--gather-root 0: to rank 0, every step).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# (GPU_MAX_HW_QUEUES is left alone: the library picks five internal streams on separate hardware queues by
# measurement, and HIP's default of 4 queues measures the same as 8 -- profiles/r6_*_bench_hw_queues_*.json.)
# dmabuf IPC: what RCCL needs between processes on this pool (the driver exports it too; a launcher that did not
# hand it down must not be what an N > 1 run fails on)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 2.4e6
D = 11
N = 65536
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _cpu_topology():
    """(model name, logical CPUs usable by this process, physical cores among them)."""
    model, phys = "unknown", set()
    try:
        usable = sorted(os.sched_getaffinity(0))
    except AttributeError:
        usable = list(range(os.cpu_count() or 1))
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
                if k == "model name" and model == "unknown":
                    model = v
            elif not line.strip() and cur:
                if int(cur.get("processor", -1)) in usable:
                    phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
    except OSError:
        pass
    return model, len(usable), (len(phys) or len(usable))


def _cpu_quota():
    """CPUs the container's cgroup lets this process use at once (cpu.max / cfs quota), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(seconds=6.0, if_filter_order=0):
    """The CPU oracle (a port of the reference path, oracle/fmd_oracle.h; what it restates:
    cFmDecoder::ProcessStream, /root/reference/src/FmDecode.cpp:417-502) timed on this host
     with NATIVE threads (oracle/fmd_oracle_bench.c: pthreads, no Python and no
    allocation in the timed loops): (a) 1 thread, 1 stereo+RDS channel; (b) one decoder per
    logical CPU, every thread its own channel state on the same 16 input blocks replayed in a
    loop.  `value` is (b), `per_core` is (a).  Where a cgroup CPU quota is in force (the GPU boxes of
    this pool: 16 CPUs of a 128-core host), more threads than the quota only get throttled -- measured
    on such a box: 798 MS/s on 16 threads, 542 MS/s on 256 -- so (b) uses as many threads as the quota
    allows and says so; `cores` is the number of threads used."""
    from oracle import oracle_py
    from tools import fmsig_py
    p = fmsig_py.default_params(FS, noise_sigma=0.01)
    blocks = np.stack([fmsig_py.generate_f32(p, b * N, N) for b in range(16)])
    params = oracle_py.FmoParams(FS, -0.15 * FS, 48000.0, 15000.0, D, 0, 0, if_filter_order, 0, 0)
    model, logical, physical = _cpu_topology()
    quota = _cpu_quota()
    threads = logical if quota is None else max(1, min(logical, int(quota + 0.5)))
    one_rate, one_calls, one_s = oracle_py.bench_threads(params, 1, seconds / 2, blocks)
    rate, calls, worst = oracle_py.bench_threads(params, threads, seconds, blocks)
    return {"value": round(rate / 1e6, 2), "unit": "MS/s", "cores": threads, "kind": "port",
            "per_core": round(one_rate / 1e6, 3), "cpu_model": model,
            "physical_cores": physical, "logical_cpus": logical,
            "cgroup_cpu_quota": quota,
            "threads": "native (pthreads), one decoder per thread; %d threads = %s"
                       % (threads, "every logical CPU" if quota is None or threads == logical
                          else "the container's cgroup CPU quota (more threads are only throttled)"),
            "scaling_vs_one_thread": round(rate / one_rate, 1),
            "sample": "%d native threads x one stereo+RDS channel at %.1f MS/s each (16 distinct "
                      "blocks of 65536 IQ replayed): %d ProcessStream calls in %.1f s; single "
                      "thread alone: %d calls in %.1f s"
                      % (threads, FS / 1e6, calls, worst, one_calls, one_s)}


def _spawn_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no launcher: this process becomes the launcher.  It has
    not imported torch and has not touched HIP (a process that has must never exec or fork workers on
    this pool); it starts N children of this same script, one per GPU, with the rendezvous variables
    torch.distributed.run would set, passes rank 0's stdout through (its JSON line stays the last line
    of stdout; the other ranks' stdout goes to stderr) and exits with the worst child's code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    base = {k: v for k, v in os.environ.items()}
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), FMD_BENCH_SPAWNED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    import signal
    import threading

    def _end_children(grace=3.0):
        """terminate(), then kill(), exactly the children started here (each is its own session:
        its process group goes with it)"""
        live = [p for p in procs if p.poll() is None]
        for p in live:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.time() + grace
        while time.time() < t_end and any(p.poll() is None for p in live):
            time.sleep(0.05)
        for p in live:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass

    def _on_signal(signum, _frame):
        # a driver's timeout, Ctrl-C, a cancelled job: the ranks must not outlive the launcher (they
        # would hold every GPU until their own watchdog fires)
        _end_children()
        raise SystemExit(128 + signum)

    old_handlers = {sg: signal.signal(sg, _on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    watchdog = 900
    for i, a in enumerate(sys.argv):
        if a == "--watchdog" and i + 1 < len(sys.argv):
            watchdog = int(sys.argv[i + 1])
        elif a.startswith("--watchdog="):
            watchdog = int(a.split("=", 1)[1])
    launcher_deadline = time.time() + watchdog + 30.0  # the ranks' own watchdog, plus a grace period
    chunks = []
    try:
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", ROLE_RANK=str(r))
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                          stderr=None, start_new_session=True))
        sys.stderr.write("bench.py: started ranks, pids %s\n" % " ".join(str(p.pid) for p in procs))
        sys.stderr.flush()
        rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        rd.start()
        # a rank that dies leaves its peers inside a rendezvous or a collective: give them a grace
        # period, then end exactly the processes started here
        deadline = None
        while any(p.poll() is None for p in procs):
            if deadline is None and any(p.poll() not in (None, 0) for p in procs):
                deadline = time.time() + 20.0
            if (deadline is not None and time.time() > deadline) or time.time() > launcher_deadline:
                _end_children()
            time.sleep(0.05)
    finally:
        _end_children(grace=1.0)  # no-op when every rank has exited
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    rcs = [p.wait() for p in procs]
    rd.join(timeout=10.0)
    out0 = b"".join(c for c in chunks if c)
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %r\n" % (bad,))
        # a signal's negative code would wrap around: report it as a plain failure
        raise SystemExit(max(rc if rc > 0 else 1 for _, rc in bad))
    raise SystemExit(0)


def _supervise_rank():
    """One rank of a run that creates an RCCL communicator (N > 1, or the forced world of one), as launched by
    torch.distributed.run or by _spawn_ranks: THIS process stays a supervisor that has not imported torch and has not
    touched HIP; the rank's work runs in a child (the same script, FMD_BENCH_WORKER=1).  The worker reports "here"
    when the host-side rendezvous has returned (every rank is present) and "up" once its RCCL communicator exists and
    a barrier behind it has passed (so "up" is all ranks or none).  RCCL's bootstrap stalls about once in 20 launches
    on this pool -- no error, every rank waits inside ncclCommInitRank --: when "up" does not follow "here" within
    FMD_BENCH_UP_TIMEOUT (60) seconds, or the worker dies before "up", the supervisor ends its worker and starts it
    again, ONCE, with a fresh rendezvous (every rank's supervisor sees the same and does the same).  The C++ loop
    has the same in tools/rank_supervisor.hpp.  Does not return."""
    import signal
    import subprocess
    import threading
    up_timeout = float(os.environ.get("FMD_BENCH_UP_TIMEOUT", "60"))
    rank = os.environ.get("RANK", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    child = [None]

    def _end(grace=2.0):
        p = child[0]
        if p is None or p.poll() is not None:
            return
        for sg, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(p.pid, sg)
            except (ProcessLookupError, PermissionError):
                pass
            t_end = time.time() + wait
            while p.poll() is None and time.time() < t_end:
                time.sleep(0.02)
            if p.poll() is not None:
                return

    def _on_signal(signum, _frame):
        _end()
        raise SystemExit(128 + signum)

    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, _on_signal)
    rc = 1
    for attempt in (0, 1):
        rfd, wfd = os.pipe()
        env = dict(os.environ, FMD_BENCH_WORKER="1", FMD_BENCH_ATTEMPT=str(attempt), FMD_BENCH_STATUS_FD=str(wfd))
        if attempt:
            # a rendezvous of its own: under torch.distributed.run the agent's store stays, the workers' keys move
            # (the prefix is the restart count); where rank 0's worker hosts the store itself, the next port
            if env.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
                env["TORCHELASTIC_RESTART_COUNT"] = str(int(env.get("TORCHELASTIC_RESTART_COUNT", "0")) + 100)
            elif "MASTER_PORT" in env:
                env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + 1)
        p = subprocess.Popen(cmd, env=env, pass_fds=(wfd,), start_new_session=True)
        child[0] = p
        os.close(wfd)
        marks = {}

        def _read(fd=rfd, marks=marks):
            with os.fdopen(fd, "r") as f:
                for line in f:
                    marks[line.strip()] = time.time()

        threading.Thread(target=_read, daemon=True).start()
        stalled = False
        while p.poll() is None and "up" not in marks:
            if "here" in marks and time.time() - marks["here"] > up_timeout:
                stalled = True
                break
            time.sleep(0.05)
        if "up" not in marks and attempt == 0 and (stalled or p.poll() not in (None, 0)):
            sys.stderr.write("bench.py: rank %s: communicator not up (%s) -- ending the worker and starting it "
                             "again, once\n" % (rank, "no answer %.0f s after the rendezvous" % up_timeout if stalled
                                                else "worker exited with %s" % p.poll()))
            sys.stderr.flush()
            _end()
            continue
        rc = p.wait()
        break
    raise SystemExit(rc if rc >= 0 else 1)


def _report(mark):
    """worker -> supervisor (_supervise_rank): the marks "here" and "up" """
    fd = os.environ.get("FMD_BENCH_STATUS_FD")
    if fd:
        try:
            os.write(int(fd), (mark + "\n").encode())
        except OSError:
            pass


def config2(args):
    """BASELINE configs[1]: 1 stereo FM channel + RDS, 2.4 MS/s, through the drop-in surface -- the call the
    reference's demux thread makes (cFmDecoder::ProcessStream, /root/reference/src/RadioReceiver.cpp:515-538):
    host IQ block in, float audio out, UECP frames through the callbacks, synchronous.  `value` = IQ samples
    per second of wall time over K calls; the JSON line also says where a call's time goes (host side:
    fmd_batch_debug_host_ms; device side: per-stage events of 24 extra calls) and what one CPU core does on
    the same blocks.  A single channel is ONE lane of work for the two serial recurrences (FM PLL, pilot PLL):
    the floor of a call is the serial stage's 5958 samples x ~107 issue slots x 4.1 cycles = 1.1 ms."""
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    from tools import fmsig_py
    pkg = load_package()
    K, W = max(args.steps, 220), args.warmup
    p = fmsig_py.default_params(FS, noise_sigma=0.005)
    nblk = 32
    blocks = [fmsig_py.generate_f32(p, b * N, N).view(np.complex64) for b in range(nblk)]
    dec = pkg.FmDecoder(FS, -0.15 * FS, 48000.0, 15000.0, D)
    view = dec.batch_view()
    for kv in args.debug_set:
        key, _, val = kv.partition("=")
        view.debug_set(key, int(val))
    for i in range(W):
        dec.ProcessStream(blocks[i % nblk])
    view.debug_host_ms()
    lat = []
    t0 = time.perf_counter()
    for i in range(W, W + K):
        t1 = time.perf_counter()
        a = dec.ProcessStream(blocks[i % nblk])
        lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    ncalls, host = view.debug_host_ms()
    lat = np.array(lat) * 1e3
    view.set_profiling(2)  # events between the stages (everything on one stream, which a single call is anyway)
    for i in range(24):
        dec.ProcessStream(blocks[i % nblk])
    stage, calls = view.stage_ms()
    view.set_profiling(0)
    stereo, frames = dec.StereoDetected(), len(dec.sink.frames.get(0, []))
    out = {
        "metric": "IQ MS/s demodulated, one decoder through cFmDecoder::ProcessStream (host buffers)",
        "value": round(N * K / dt / 1e6, 2), "unit": "MS/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 1 stereo FM channel + RDS @2.4 MS/s, 65536 IQ per call, "
                               "fmd_process_stream (host IQ in, host audio out, UECP callbacks), %d calls" % K,
                   "stereo_detected": bool(stereo), "uecp_frames": frames,
                   "debug_set": args.debug_set},
        "latency_ms": {"mean": round(float(lat.mean()), 4), "p50": round(float(np.percentile(lat, 50)), 4),
                       "p99": round(float(np.percentile(lat, 99)), 4), "max": round(float(lat.max()), 4),
                       "host_side_mean": {k: round(v, 4) for k, v in host.items()},
                       "host_side_note": "copy_in: IQ block to the device; submit: the call's ~25 launches; "
                                         "wait_copy_out: waiting for them + audio back; rds_callbacks: group "
                                         "collection + UECP group decoder",
                       "device_stage_ms": {k: round(v, 4) for k, v in stage.items() if v >= 0},
                       "device_stage_calls": calls,
                       "floor": "serial stage: 5958 baseband samples x 106.5 issue slots x 4.1 cycles at 2.39 GHz "
                                "= 1.09 ms for the FM wave alone (one channel = one lane of a strictly serial "
                                "recurrence)"},
    }
    if not args.no_cpu_baseline:
        cb = cpu_baseline(seconds=4.0)
        out["cpu_baseline"] = cb
        out["cpu_baseline"]["note"] = "per_core (one thread, one channel) is the figure that compares with `value`"
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 240 steps = 0.6 s: the pipeline's fill and drain (three calls deep, ~5 ms) stay below 1 % of the
    # timed region (with 24 steps they were 8 %)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--channels", type=int, default=8192, help="channels per GPU")
    ap.add_argument("--ring", type=int, default=10, help="distinct input blocks resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None,
                    help="development aid: another build of libfmd_hip.so (A/B of compile-time constants)")
    ap.add_argument("--captures", type=int, default=1,
                    help="config3 only: G captures x 256 stations each in one batch (config 3 scaled out until it "
                         "fills the chip; fmd_batch_set_channels_per_capture)")
    ap.add_argument("--workload", default="config4", choices=["config4", "config3", "config5", "config2"],
                    help="config4 (default, the metric's workload): independent channels @2.4 MS/s; "
                         "config2: ONE stereo+RDS decoder through the cFmDecoder surface (fmd_process_stream, host "
                         "buffers in and out, callbacks): per-call latency and where it goes; "
                         "config3: 256 channels from ONE shared capture (table_size 256); "
                         "config5: 4096-tap IF FIR @10 MS/s, D=46, 4096 channels")
    ap.add_argument("--input", default="f32", choices=["f32", "u8"],
                    help="f32: complex<float> blocks, the ProcessStream argument (BASELINE metric); "
                         "u8: RTL-SDR byte pairs converted inside the IF kernel (SURVEY 8(f)-2, "
                         "reported as its own workload: 2 B instead of 8 B read per IQ sample)")
    ap.add_argument("--concurrency", type=int, default=2, choices=[0, 1, 2],
                    help="fmd_batch_set_concurrency mode (2 = calls overlap, the default)")
    ap.add_argument("--stage-profile", action="store_true",
                    help="after the timed region, run 3 extra steps with per-stage events")
    ap.add_argument("--lag", type=int, default=3, choices=[1, 2, 3, 4],
                    help="steps between submitting a call and consuming its outputs (host never blocks "
                         "on a call younger than this)")
    ap.add_argument("--fir-reduction", type=int, default=0, choices=[0, 1, 2],
                    help="0: sequential tap order, bit-exact (default, what every reported figure uses); "
                         "1: opt-in shuffle-reduced tap sum; 2: fused multiply-add in the reference's order (both "
                         "not bit-exact: parity waived, measured for docs/MEASUREMENTS.md only)")
    ap.add_argument("--verify", action="store_true",
                    help="before the timed region: rank 0 checks the audio and RDS records it gathered "
                         "from every rank (a few channels each, first steps) bit for bit against its own "
                         "recomputation of those channels in a small batch; exits 4 on a mismatch.  "
                         "On by default with more than one rank (a multi-GPU number is only reported for "
                         "a gather that was checked); --no-verify turns it off")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--gather-root", default="rotate", choices=["0", "rotate"],
                    help="N > 1: which rank receives a step's outputs -- rotate (default): step i goes to rank i %% N "
                         "(fmd_gather_step_root), every rank takes 1 / N of the receive load; 0: rank 0, every step -- it "
                         "alone then takes 7 x 88 MB of writes per step at 8 GPUs, 9 %% of its throughput, and the node "
                         "runs at its pace (emulated on one GPU: docs/MEASUREMENTS.md, round 6)")
    ap.add_argument("--emulate-peers", type=int, default=0, metavar="P",
                    help="sizing of rank 0 on one GPU (with FMD_BENCH_FORCE_DIST=1, a world of one): every step's gather "
                         "also writes what P more ranks' receives would write into rank 0's buffers "
                         "(fmd_gather_debug_emulate_peers)")
    ap.add_argument("--emulate-wgs", type=int, default=2, help="workgroups per emulated peer")
    ap.add_argument("--emulate-role", default="root", choices=["root", "sender", "rotate"],
                    help="which rank of a P + 1 rank node this GPU plays: the root of every step (rank 0 of the "
                         "fixed-root gather), a sender in every step (its message read once per step), or a rank of a "
                         "rotating root (receives in every (P+1)-th step, sends otherwise)")
    ap.add_argument("--emulate-peer-channels", type=int, default=0,
                    help="channels each emulated peer's message is sized for (default: --channels); rank 0 of an "
                         "unequal split decodes fewer channels than every peer sends")
    ap.add_argument("--no-host-throttle", action="store_true",
                    help="N > 1 / --verify path: do not block the host on the call LAG steps back (round 5's behaviour: "
                         "the host runs eight calls ahead)")
    ap.add_argument("--side-stream", action="store_true",
                    help="development: submit from a non-default torch stream (the null stream orders itself "
                         "against every blocking stream of the process)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE",
                    help="development switch of the batch (fmd_batch_debug_set), e.g. resampler=0; "
                         "recorded in the JSON line; the reported figures use none")
    ap.add_argument("--watchdog", type=int, default=900,
                    help="seconds after which a run that has not finished kills itself (a hung "
                         "collective or kernel must not keep the box busy)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.workload == "config2":
        return config2(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(args.gpus)  # does not return
    if ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("FMD_BENCH_FORCE_DIST") == "1")
            and os.environ.get("FMD_BENCH_WORKER") != "1" and os.environ.get("FMD_BENCH_BACKEND", "nccl") == "nccl"):
        _supervise_rank()  # does not return: the rank's work runs in a child that can be started again
    if os.environ.get("FMD_BENCH_TEST_SUPERVISOR") == "1":  # test aid (tests/test_bench_launcher.py): a worker
        _report("here")                                     # whose communicator never comes up the first time
        if os.environ.get("FMD_BENCH_ATTEMPT") == "0":
            time.sleep(600)
        _report("up")
        print(json.dumps({"test_worker": True, "attempt": int(os.environ["FMD_BENCH_ATTEMPT"]),
                          "rank": int(os.environ.get("RANK", "0")),
                          "restart_count": os.environ.get("TORCHELASTIC_RESTART_COUNT"),
                          "master_port": os.environ.get("MASTER_PORT")}), flush=True)
        return
    if os.environ.get("FMD_BENCH_TEST_HANG") == "1" and os.environ.get("FMD_BENCH_SPAWNED") == "1":
        time.sleep(600)  # test aid (tests/test_bench_launcher.py): a rank that never finishes

    import threading

    def _expired():
        sys.stderr.write("bench.py: watchdog expired after %d s, exiting\n" % args.watchdog)
        sys.stderr.flush()
        os._exit(3)

    wd = threading.Timer(args.watchdog, _expired)
    wd.daemon = True
    wd.start()
    if os.environ.get("FMD_BENCH_WORKER") == "1":
        # a worker must not outlive its supervisor (_supervise_rank: a launcher that is killed outright cannot end it)
        parent = os.getppid()

        def _orphan_check():
            while os.getppid() == parent:
                time.sleep(1.0)
            os._exit(3)

        threading.Thread(target=_orphan_check, daemon=True).start()

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    from tools import fmsig_py

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (either let bench.py start its own ranks: no "
                         "WORLD_SIZE in the environment, or launch it with torch.distributed.run)"
                         % (args.gpus, world))
    if os.environ.get("FMD_BENCH_SHARE_GPU") == "1":
        local_rank = 0  # development aid, see below
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.side_stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    # RCCL ("nccl") on a real multi-GPU node.  FMD_BENCH_BACKEND=gloo + FMD_BENCH_SHARE_GPU=1 is a
    # development aid: several ranks on ONE GPU with a host-staged gather, to exercise the N > 1
    # control flow where only one GPU exists.
    backend = os.environ.get("FMD_BENCH_BACKEND", "nccl")
    # FMD_BENCH_FORCE_DIST=1: run the N > 1 code path (communicator, side stream, gather to rank 0,
    # group counting from the gathered records) with a world of ONE rank -- what a box with a single
    # GPU can exercise of the RCCL path: the communicator is initialised and every gather call is made.
    dist_on = world > 1 or os.environ.get("FMD_BENCH_FORCE_DIST") == "1"
    emu_peers = args.emulate_peers if (dist_on and world == 1) else 0
    if args.emulate_peers and not emu_peers:
        raise SystemExit("--emulate-peers needs FMD_BENCH_FORCE_DIST=1 and --gpus 1 (a world of one)")
    if world > 1 and not args.no_verify:
        args.verify = True
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29549")
    # torch.distributed only carries the rendezvous and the barriers, over gloo on the host: the process then holds
    # ONE RCCL communicator (the C++ gather's) instead of two (torch's nccl process group has its own, with its
    # streams and proxy thread) -- half as many bootstraps that can stall, and the host-side rendezvous returns when
    # every rank is present, which is where the supervisor's clock starts (_supervise_rank).
    # FMD_BENCH_RENDEZVOUS=nccl brings torch's own communicator back (round 5's default).
    pg_backend = os.environ.get("FMD_BENCH_RENDEZVOUS", "gloo" if backend == "nccl" else backend)
    pg_dev = dev if pg_backend == "nccl" else "cpu"
    if dist_on:
        if pg_backend == "nccl":
            _report("here")
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=pg_backend, rank=rank, world_size=world)
            _report("here")

    global FS, D
    order = 0
    table = 0
    shared = False
    if args.workload == "config5":
        FS, D, order = 10e6, 46, 4096
        if os.environ.get("FMD_BENCH_GEOM"):  # dev aid: "fs,D,order" -- the other long-filter window layouts
            g = os.environ["FMD_BENCH_GEOM"].split(",")
            FS, D, order = float(g[0]), int(g[1]), int(g[2])
        if args.channels == 8192:
            args.channels = 4096
    elif args.workload == "config3":
        table, shared = 256, True
        if args.channels == 8192:
            args.channels = 256 * max(1, args.captures)
    u8 = args.input == "u8"
    if u8 and shared:
        raise SystemExit("--input u8 is implemented for the per-channel workloads")
    in_dtype = torch.uint8 if u8 else torch.float32
    pkg = load_package()
    if args.lib:
        pkg.LIB_PATH = os.path.abspath(args.lib)
    import importlib
    dg = importlib.import_module(pkg.__name__ + ".dist_gather")
    C = args.channels
    K, W = args.steps, args.warmup
    ring = max(1, min(args.ring, K + W))

    # ---- inputs: ring of blocks, [ring][C][N] complex64, generated on the device ----
    if shared:
        # one capture = six stations 200-300 kHz apart, generated per station and summed
        offs = (-600e3, -360e3, -150e3, 75e3, 300e3, 600e3)
        st = [fmsig_py.default_params(FS, f_offset=f0, amp=0.12, noise_sigma=0.004, seed=50 + i,
                                      pi=0x5000 + i, ps="CAP%05d" % i, f_left=500.0 + 300 * i)
              for i, f0 in enumerate(offs)]
        G = max(1, args.captures)
        if C % G:
            raise SystemExit("--captures must divide --channels")
        tmp = torch.empty((len(st), N, 2), dtype=torch.float32, device=dev)
        iq = torch.empty((ring, G, N, 2), dtype=torch.float32, device=dev)
        for g in range(G):  # every capture its own six stations (other seeds, tones and PI codes)
            stg = [fmsig_py.default_params(FS, f_offset=f0, amp=0.12, noise_sigma=0.004, seed=50 + i + 16 * g,
                                           pi=0x5000 + i + 16 * g, ps="CAP%05d" % (i + 16 * g),
                                           f_left=500.0 + 300 * i + 7 * g)
                   for i, f0 in enumerate(offs)] if g else st
            gen = fmsig_py.DeviceGenerator(stg, dev)
            for r in range(ring):
                gen.generate(tmp, r * N, N)
                iq[r, g] = tmp.sum(dim=0)
    else:
        chans = [fmsig_py.channel_params(FS, rank * C + c) for c in range(C)]
        gen = fmsig_py.DeviceGenerator(chans, dev)
        iq = torch.empty((ring, C, N, 2), dtype=in_dtype, device=dev)
        for r in range(ring):
            gen.generate(iq[r], r * N, N)
    torch.cuda.synchronize()

    # the communicator (and its streams) first: the decoder picks its internal streams by a probe
    # when the batch is created and should see everything else that uses hardware queues
    if dist_on:
        dg.gather_preflight(pg_dev)
    comm_stream = torch.cuda.Stream(device=dev) if dist_on else None
    # development aid: other users of hardware queues in the process, created first
    extra_streams = [torch.cuda.Stream(device=dev)
                     for _ in range(int(os.environ.get("FMD_BENCH_EXTRA_STREAMS", "0")))]
    G = max(1, args.captures) if shared else 1
    cpc = C // G  # channels per capture
    shifts = (np.arange(C, dtype=np.int32) % table) - table // 2 if shared else None
    batch = pkg.Batch(pkg.make_params(FS, 0.0 if shared else -0.15 * FS, 48000.0, 15000.0, D,
                                      table_size=table, if_filter_order=order,
                                      fir_reduction={0: 0, 1: 0x101, 2: 0x102}[args.fir_reduction]),
                      C, tuning_shifts=shifts, device=local_rank, record_callbacks=False)
    if G > 1:
        batch.set_channels_per_capture(cpc)
    for kv in args.debug_set:
        key, _, val = kv.partition("=")
        batch.debug_set(key, int(val))
    a_stride = (batch.max_audio_floats(N) + 63) // 64 * 64
    NBUF = args.lag + 3  # outputs are consumed LAG steps after they are produced, then gathered
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device=dev) for _ in range(NBUF)]
    RCAP = C  # RDS records per rank per step (a group takes 87.6 ms, a step 27.3 ms: <= 1 per channel)
    CMSG = max(C, args.emulate_peer_channels) if emu_peers else C  # channels a rank's gather message is sized for
    rds_dev = [torch.zeros((RCAP, 4), dtype=torch.int32, device=dev) for _ in range(NBUF)]
    # N > 1 (and --verify): the RDS groups of a step leave the decoder as fixed-size records in device
    # memory (fmd_batch_export_rds_device) and go into the gather as they are -- no host round trip.
    # N = 1: the host pulls them (fmd_batch_collect_rds), like an application that runs the UECP
    # group decoder would.
    use_export = dist_on or args.verify
    g_audio = g_rds = None
    rotate = args.gather_root == "rotate" and dist_on

    def root_of(i):
        return i % world if rotate else 0

    audio_own, rds_own = audio, rds_dev  # (a root's outputs are produced in place: see below)
    if (rank == 0 or rotate) and (dist_on or args.verify):
        on = dev if (backend == "nccl" or not dist_on) else "cpu"
        # one tensor per slot, [world][...]: g_audio[slot][r] is rank r's part (the C++ gather writes at
        # rank * size; torch.distributed.gather takes the list of the parts)
        # (--emulate-peers P, world of one: room for the P ranks whose receives are emulated)
        # (--emulate-peer-channels: the emulated peers' messages are sized for that many channels each -- rank 0 of an
        # unequal split decodes fewer channels than it receives per peer; its own part of the buffers is then larger
        # than what it fills)
        g_audio = [torch.empty((world + emu_peers, CMSG, a_stride), dtype=torch.float32, device=on) for _ in range(NBUF)]
        g_rds = [torch.zeros((world + emu_peers, CMSG, 4), dtype=torch.int32, device=on) for _ in range(NBUF)]
        if on == dev and backend == "nccl" and dist_on:
            # rank 0 has its own outputs produced IN PLACE, in its part of the receive buffers: the gather then has
            # nothing to copy for it (fmd_gather_step: d_audio == d_all_audio), 88 MB per step less through HBM
            # (a rotating root: rank r's part is the r-th, and it is used in the steps r is the root of)
            audio = [g[rank][:C] for g in g_audio]
            rds_dev = [g[rank][:RCAP] for g in g_rds]
            for t in rds_dev:
                t.zero_()
    # RCCL: the data path is the C++ gather of include/fmd_gather.h (grouped ncclSend / ncclRecv on a
    # stream of its own); its communicator's id travels over the torch.distributed rendezvous
    gth = None
    if dist_on and backend == "nccl":
        gmod = importlib.import_module(pkg.__name__ + ".gather")
        uid = torch.zeros(gmod.ID_BYTES, dtype=torch.uint8, device=pg_dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(gmod.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        gth = gmod.Gather(bytes(uid.cpu().numpy().tobytes()), rank, world, local_rank, CMSG * a_stride, CMSG)
        dist.barrier()  # every rank's communicator is up, or none says so
        _report("up")
        if emu_peers:
            gth.emulate_peers(emu_peers, args.emulate_wgs)
            gth.emulate_role({"root": 1, "sender": 0, "rotate": emu_peers + 1}[args.emulate_role])
    group_acc = torch.zeros((), dtype=torch.int64, device=dev)  # groups counted on the device
    stream = torch.cuda.current_stream().cuda_stream
    pending = [None] * NBUF
    gather_events = []  # (start, stop) on the side stream around every step's gather calls
    total_groups = 0

    batch.set_concurrency(args.concurrency)  # 2: FIR of step i+1 overlaps the serial stages of step i
    # outputs of step i are consumed after step i+LAG is submitted: the host blocks on nothing younger
    # (3 and 4 measure the same, 244 GS/s on one box: the period is set by the device's work)
    LAG = args.lag
    state = {"submitted": -1, "finalized": -1}

    def pull_groups(lag):
        nonlocal total_groups
        got = batch.collect_rds_array(cap=4 * RCAP, stream=stream, lag=lag)
        total_groups += int(got.size)

    def finalize(i, lag):
        """Outputs of step i (call index i+1), complete on the torch stream: with N > 1 its audio and
        RDS records are gathered to the step's root over RCCL on the side stream (overlapping the next steps'
        compute); rank 0 counts the groups that arrived."""
        slot = i % NBUF
        if gth is not None:
            # export of the RDS records (on the torch stream) + the step's sends / receives (on the
            # library's stream, behind the torch stream as it stands now): one C call
            root = root_of(i)
            a_buf, r_buf = out_bufs(i)
            pending[slot] = ("ticket", gth.step(batch, lag, rank * C, a_buf.data_ptr(), r_buf.data_ptr(),
                                                g_audio[slot].data_ptr() if rank == root else None,
                                                g_rds[slot].data_ptr() if rank == root else None, stream, root=root),
                             root)
        elif use_export:
            batch.export_rds_device(out_bufs(i)[1].data_ptr(), RCAP, channel_offset=rank * C,
                                    stream=stream, lag=lag)
        if dist_on and gth is None:
            ev = torch.cuda.Event()
            ev.record()
            if True:  # host-staged over gloo (development aid: several ranks on one GPU)
                tg0 = time.perf_counter()
                ev.synchronize()
                root = root_of(i)  # (a rotating root over gloo too: the same host logic as with RCCL, two ranks on one GPU)
                a_buf, r_buf = out_bufs(i)
                a_h, r_h = a_buf.cpu(), r_buf.cpu()
                w = dg.gather_step(a_h, r_h, list(g_audio[slot]) if rank == root else None,
                                   list(g_rds[slot]) if rank == root else None, dst=root, async_op=True)
                pending[slot] = [root] + list(w)
                host_t["gather_host"] = host_t.get("gather_host", 0.0) + (time.perf_counter() - tg0)
        elif use_export and gth is None:  # one rank, --verify: "gathered" = this rank's own outputs
            g_audio[slot][0].copy_(audio[slot], non_blocking=True)
            g_rds[slot][0].copy_(rds_dev[slot], non_blocking=True)
            group_acc.add_((rds_dev[slot][:, 0] != 0).sum())
        state["finalized"] = i

    def out_bufs(i):
        """Where step i's audio and RDS records are produced: in this rank's part of its receive buffers in the steps
        it is the root of (the gather then copies nothing for it), in buffers of its own otherwise."""
        slot = i % NBUF
        if rotate and rank != root_of(i):
            return audio_own[slot], rds_own[slot]
        return audio[slot], rds_dev[slot]

    def release(slot):
        """Before a slot's buffers are written again: its gather must have read them."""
        if pending[slot] is None:
            return
        if isinstance(pending[slot], tuple):  # the C++ gather: order the torch stream behind that step
            gth.wait_for(pending[slot][1], stream)
            if rank == pending[slot][2]:  # the step's root counts what arrived
                group_acc.add_((g_rds[slot][:, :, 0] != 0).sum())
        elif isinstance(pending[slot], list):
            for w in pending[slot][1:]:
                w.wait()
            if rank == pending[slot][0]:  # the step's root counts what arrived
                group_acc.add_(int((g_rds[slot][:, :, 0] != 0).sum()))
        else:
            torch.cuda.current_stream().wait_event(pending[slot])  # device-side wait
        pending[slot] = None

    host_t = {"process": 0.0, "collect": 0.0}
    # dev aid (FMD_BENCH_STEPTIMES=1): an event on the torch stream behind every call's completion
    step_events = [] if os.environ.get("FMD_BENCH_STEPTIMES") else None

    def step(i):
        slot = i % NBUF
        release(slot)
        th0 = time.perf_counter()
        nf = batch.process_device(iq[i % ring].data_ptr(), (N if G > 1 else 0) if shared else N, N,
                                  out_bufs(i)[0].data_ptr(), a_stride, stream, u8=u8)
        host_t["process"] += time.perf_counter() - th0
        state["submitted"] = i
        if i - LAG > state["finalized"]:
            # orders the torch stream after the calls that are at least LAG old, drains their groups
            th0 = time.perf_counter()
            batch.wait(stream=stream, lag=LAG)
            th1 = time.perf_counter()
            if not use_export:
                pull_groups(LAG)
            host_t["collect"] += time.perf_counter() - th1
            host_t["wait"] = host_t.get("wait", 0.0) + (th1 - th0)
            while state["finalized"] < i - LAG:
                finalize(state["finalized"] + 1, LAG)
                if step_events is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    step_events.append((state["finalized"], e))
            if use_export and not args.no_host_throttle:
                # The host blocks on the call LAG steps back, like the N = 1 path does inside fmd_batch_collect_rds:
                # without it nothing stops the host before the decoder's own limit of eight calls in flight, and the
                # device runs 15-20 % slower behind a queue that deep at the driver's flags (round 6: 225 500 -> 251 500 MS/s
                # with a world of one; the gather's own stream is not waited for)
                th2 = time.perf_counter()
                torch.cuda.current_stream().synchronize()
                host_t["wait"] = host_t.get("wait", 0.0) + (time.perf_counter() - th2)
        return nf

    def drain():
        # the last LAG calls one by one, so that every call's groups land in its own record buffer
        while state["finalized"] < state["submitted"]:
            lag = state["submitted"] - (state["finalized"] + 1)
            batch.wait(stream=stream, lag=lag)
            if not use_export:
                pull_groups(lag)
            finalize(state["finalized"] + 1, lag)
            if step_events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                step_events.append((state["finalized"], e))
        batch.wait(stream=stream)
        for slot in range(NBUF):
            release(slot)
        if comm_stream is not None:
            comm_stream.synchronize()
        torch.cuda.synchronize()

    def barrier():
        if dist_on:
            dist.barrier()

    def reduce_scalar(x, op):
        t = torch.tensor([x], dtype=torch.float64, device=pg_dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    # ---- --verify: what rank 0 received is what a single-rank recomputation gives ----
    verify = None
    base = 0
    if args.verify:
        V = min(4, NBUF, ring)
        nfs = [step(i) for i in range(V)]
        drain()
        base = V
        ok = 1.0
        if rank == 0 or rotate:  # (a rotating root: every rank checks the steps it received)
            picks = sorted({0, 1, C // 2, C - 1})
            if shared:
                vshifts = np.concatenate([shifts[picks] for _ in range(world)]).astype(np.int32)
                vgen = None
            else:
                vshifts = None
                vgen = fmsig_py.DeviceGenerator(
                    [fmsig_py.channel_params(FS, r * C + c) for r in range(world) for c in picks], dev)
            nv = world * len(picks)
            vb = pkg.Batch(pkg.make_params(FS, 0.0 if shared else -0.15 * FS, 48000.0, 15000.0, D,
                                           table_size=table, if_filter_order=order),
                           nv, tuning_shifts=vshifts, device=local_rank, record_callbacks=False)
            viq = torch.empty((nv, N, 2), dtype=in_dtype, device=dev)
            vaudio = torch.zeros((nv, a_stride), dtype=torch.float32, device=dev)
            bad = []
            for i in range(V):
                if shared and G > 1:  # the verify batch takes a row per channel: each pick's own capture
                    for j in range(nv):
                        viq[j].copy_(iq[i % ring][picks[j % len(picks)] // cpc])
                    src, vstride = viq, N
                elif shared:
                    src, vstride = iq[i % ring], 0
                else:
                    vgen.generate(viq, (i % ring) * N, N)
                    src, vstride = viq, N
                vnf = vb.process_device(src.data_ptr(), vstride, N, vaudio.data_ptr(), a_stride, stream,
                                        u8=u8)
                vb.wait(stream=stream)
                vg = vb.collect_rds_array(cap=4 * nv, stream=stream)
                torch.cuda.synchronize()
                if root_of(i) != rank:
                    continue
                want_a = vaudio[:, :vnf].cpu().numpy().view(np.uint32)
                slot = i % NBUF
                for r in range(world):
                    got_a = g_audio[slot][r][picks, :vnf].cpu().numpy().view(np.uint32)
                    if vnf != nfs[i] or not np.array_equal(got_a, want_a[r * len(picks):(r + 1) * len(picks)]):
                        bad.append(("audio", i, r))
                    got_g = sorted(x for x in dg.unpack_rds_records(g_rds[slot][r].cpu().numpy())
                                   if x[0] - r * C in picks)
                    want_g = sorted((r * C + picks[int(ch) - r * len(picks)], int(ci),
                                     tuple(int(v) for v in bl))
                                    for ch, ci, bl in zip(vg["channel"], vg["call_index"], vg["blocks"])
                                    if r * len(picks) <= ch < (r + 1) * len(picks))
                    if got_g != want_g:
                        bad.append(("rds", i, r))
            vb.close()
            verify = {"steps": V, "channels_per_rank": picks, "ranks": world, "mismatches": bad,
                      "per_rank_ok": [not any(b[2] == r for b in bad) for r in range(world)],
                      "ok": not bad, "gather_root": args.gather_root}
            ok = 0.0 if bad else 1.0
            if bad:
                sys.stderr.write("bench.py --verify: MISMATCH %r\n" % (bad,))
        if dist_on:
            ok = reduce_scalar(ok, dist.ReduceOp.MIN)
            if rotate:  # the roots' verdicts on every sender, combined
                per = [reduce_scalar(1.0 if verify["per_rank_ok"][r] else 0.0, dist.ReduceOp.MIN) for r in range(world)]
                verify["per_rank_ok"] = [p == 1.0 for p in per]
                verify["ok"] = ok == 1.0
        if ok != 1.0:
            raise SystemExit(4)

    for i in range(base, base + W):
        step(i)
    drain()
    host_t["process"] = host_t["collect"] = host_t["wait"] = host_t["gather_host"] = 0.0
    gather_events.clear()
    if gth is not None:
        gth.ms_per_step()  # forget the warm-up's
    batch.set_profiling(1)  # HIP events around the IF FIR kernel of every timed call
    total_groups = 0
    group_acc.zero_()
    barrier()
    torch.cuda.synchronize()
    if step_events is not None:
        step_events.clear()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    t0 = time.perf_counter()
    for i in range(base + W, base + W + K):
        nf = step(i)
    drain()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if step_events is not None and rank == 0:
        ts = [e0.elapsed_time(e) for _, e in step_events]
        sys.stderr.write("STEPTIMES wall %.3f ms; call completion (ms since t0): %s\n  intervals: %s\n" % (
            dt * 1e3, " ".join("%.2f" % t for t in ts),
            " ".join("%.2f" % (b - a) for a, b in zip([0.0] + ts[:-1], ts))))
    if use_export:
        total_groups = int(group_acc.item())  # the root: groups that arrived from every rank
        if rotate:  # every rank was the root of every world-th step
            total_groups = int(reduce_scalar(float(total_groups), dist.ReduceOp.SUM))
    serial_probe = None
    if "serial_probe=1" in args.debug_set:  # dev aid: per-workgroup timing of the serial stage
        pr = batch.debug_serial_probe()
        where = pr[:, :, 2] >> 40          # CU/SH/SE byte of HW_ID, XCC id above it
        pr[:, :, 2] &= (1 << 40) - 1
        serial_probe = []
        if os.environ.get("FMD_PROBE_RAW"):  # per launch: the slow workgroups (index: xcc.se.cu)
            order = np.argsort(pr[:, 0, 0])
            for l in order:
                u = pr[l]
                if not (u[:, 1] > 0).any():
                    continue
                cyc = u[:, 2] / 1e6
                w = where[l]
                thr = cyc[u[:, 1] > 0].min() * 1.05
                slow = [i for i in range(len(u)) if u[i, 1] > 0 and cyc[i] > thr]
                print("PROBE_RAW launch", int(l), "min Mcyc", round(float(cyc[u[:, 1] > 0].min()), 3), "slow:",
                      " ".join("%d:%d.%d.%d" % (i, int(w[i] >> 8), int(w[i] & 0xff) >> 5, int(w[i] & 0xf))
                               for i in slow), file=sys.stderr)
        for l in range(pr.shape[0]):
            u = pr[l][pr[l][:, 1] > 0]
            if len(u):
                d_us = (u[:, 1] - u[:, 0]) / 100.0
                serial_probe.append({"start": int(u[:, 0].min()), "workgroups": int(len(u)),
                                     "start_skew_us": round(float(u[:, 0].max() - u[:, 0].min()) / 100.0, 1),
                                     "span_us": round(float(u[:, 1].max() - u[:, 0].min()) / 100.0, 1),
                                     "wg_us_mean": round(float(d_us.mean()), 1),
                                     "wg_us_min": round(float(d_us.min()), 1),
                                     "wg_us_max": round(float(d_us.max()), 1),
                                     "mhz": round(float((u[:, 2] / d_us).mean()), 0)})
        serial_probe.sort(key=lambda r: r["start"])
        t00 = serial_probe[0]["start"] if serial_probe else 0
        for r in serial_probe:
            r["start"] = round((r["start"] - t00) / 100.0, 1)
    dt_own = dt
    if dist_on:
        dt = reduce_scalar(dt, dist.ReduceOp.MAX)
    stage, calls = batch.stage_ms()
    fir_ms = stage["if_fir"]
    if os.environ.get("FMD_BENCH_TIMELINE") and rank == 0:  # dev aid: what fill and drain are made of
        tl = batch.debug_timeline()
        sys.stderr.write("TIMELINE (ms since the first timed call's FIR start; wall %.3f ms)\n"
                         "call  fir_start fir_end  ser_start ser_end  tail_start tail_end  hbchain_start hbchain_end  resample_start resample_end\n" % (dt * 1e3))
        for c, row in enumerate(tl):
            sys.stderr.write("%4d  %s\n" % (c, "  ".join("%8.3f" % v for v in row)))
    # what makes an N > 1 run explain itself: every rank's own time per step and FIR time, and what the
    # gather costs on the side stream (on rank 0: receiving from every peer; elsewhere: sending)
    gather_ms = None
    if gth is not None:
        gather_ms = gth.ms_per_step()
    elif gather_events:
        torch.cuda.synchronize()
        gather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / len(gather_events)
    elif host_t.get("gather_host"):
        gather_ms = host_t["gather_host"] / K * 1e3
    per_rank = None
    if dist_on:
        mine = torch.tensor([dt_own / K * 1e3, fir_ms, -1.0 if gather_ms is None else gather_ms],
                            dtype=torch.float64, device=pg_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "ms_per_step": round(float(t[0]), 4), "if_fir_ms": round(float(t[1]), 4),
                     "gather_ms_per_step": None if float(t[2]) < 0 else round(float(t[2]), 4)}
                    for r, t in enumerate(allr)]
        if gth is not None:
            # what every rank's OWN communicator says about itself (ncclCommCount / ncclCommUserRank of the
            # gather's communicator, include/fmd_gather.h): N processes that each ran a world of one would
            # show ranks_seen 1 here
            gi = gth.info()
            mine_i = torch.tensor([gi["ranks_seen"], gi["rank"], gi["device"], gi["steps_issued"]],
                                  dtype=torch.int64, device=pg_dev)
            alli = [torch.zeros_like(mine_i) for _ in range(world)]
            dist.all_gather(alli, mine_i)
            for r, t in enumerate(alli):
                per_rank[r].update({"rccl_ranks_seen": int(t[0]), "rccl_rank": int(t[1]), "rccl_device": int(t[2]),
                                    "gather_steps_issued": int(t[3])})
    host_ms = {k: v / K * 1e3 for k, v in host_t.items()}  # the timed region's, before the extra steps

    # after the timed region: four more steps with the stages one after the other on one stream and
    # events between them -- every kernel alone on the chip (the FIR's figure alone goes beside the
    # in-pipeline one in `roofline`)
    stage_all = None
    if args.stage_profile or args.concurrency == 2:
        drain()
        if args.concurrency != 0:
            batch.set_concurrency(0)
        batch.set_profiling(2)
        for i in range(base + W + K, base + W + K + 4):
            step(i)
        drain()
        stage_all, _ = batch.stage_ms()
    batch.set_profiling(0)

    # What the host-buffer boundary (fmd_batch_process_host: ProcessStream's own arguments are host
    # pointers) could sustain at best: every IQ byte crosses PCIe once.  Measured here, never `value`.
    pcie = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        hb = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
        db = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        db.copy_(hb, non_blocking=True)
        torch.cuda.synchronize()
        tpc = time.perf_counter()
        for _ in range(4):
            db.copy_(hb, non_blocking=True)
        torch.cuda.synchronize()
        gbs = 4 * hb.numel() / (time.perf_counter() - tpc) / 1e9
        in_b = 2.0 if u8 else 8.0
        pcie = {"h2d_GBps_pinned": round(gbs, 1), "value": round(gbs * 1e9 / in_b / 1e6, 1), "unit": "MS/s",
                "note": "upper bound with host-resident input: %.0f B per IQ sample over the measured pinned "
                        "host-to-device rate, copies fully overlapped with compute; `value` above is with the "
                        "input resident in HBM (SURVEY 8(d))" % in_b}
        del hb, db
    if rank == 0:
        samples_per_step = C * N
        value = world * samples_per_step * K / dt / 1e6
        # algorithmic bytes of the fused tuner+FIR kernel: read 8 B per IQ sample (8/C for a capture
        # shared by C channels), write 8/D B; one launch processes C*N samples.
        in_bytes = 2.0 if u8 else 8.0
        bytes_per_launch = samples_per_step * ((in_bytes / cpc if shared else in_bytes) + 8.0 / D)
        taps = order if order else 8 * D
        flops_per_launch = samples_per_step * (6.0 + 4.0 * taps / D)
        achieved = bytes_per_launch / (fir_ms * 1e-3) / 1e9
        out = {
            "metric": "IQ MS/s demodulated (whole node) + achieved HBM GB/s on FIR stage",
            "value": round(value, 1), "unit": "MS/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {
                "config4": "BASELINE configs[3] per-GPU shard: %d independent FM stereo+RDS "
                           "channels/GPU @2.4 MS/s, 65536 IQ/channel/step, D=11, 88-tap IF FIR, "
                           "full ProcessStream path" % C,
                "config3": ("BASELINE configs[2]: %d channels freq-shifted from ONE shared 2.4 MS/s "
                            "capture (table_size 256), full ProcessStream path" % C) if G == 1 else
                           ("BASELINE configs[2] scaled out: %d captures x %d channels freq-shifted from each "
                            "(2.4 MS/s, table_size 256), full ProcessStream path" % (G, cpc)),
                "config5": ("BASELINE configs[4]: %d channels @10 MS/s, D=46, 4096-tap IF FIR, full "
                            "ProcessStream path" % C) if not os.environ.get("FMD_BENCH_GEOM") else
                           ("dev geometry %d channels @%.3g MS/s, D=%d, %d-tap IF FIR" % (C, FS / 1e6, D, order))}[
                               args.workload]
                + (" -- input as RTL-SDR u8 byte pairs, ReadAsyncCB conversion fused into the IF "
                   "kernel (SURVEY 8(f)-2; not the BASELINE metric's input format)" if u8 else ""),
                       "input_format": args.input,
                       "fir_reduction": {0: "sequential (bit-exact)", 1: "shuffle (opt-in, NOT bit-exact)",
                                         2: "fused multiply-add (opt-in, NOT bit-exact)"}[args.fir_reduction],
                       "channels_per_gpu": C, "samples_per_call": N, "input_ring_blocks": ring,
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "internal_streams_sharing_a_hw_queue": batch.streams_sharing_queue(),
                       "audio_floats_per_channel_step": nf, "rds_groups_in_timed_region": total_groups,
                       "gather": ("gather of audio + RDS records per step over %s (%d rank%s)"
                                  % ("RCCL, grouped ncclSend / ncclRecv from C++ (include/fmd_gather.h)"
                                     if backend == "nccl" else backend, world, "" if world == 1 else "s")
                                  + ("; the root rotates: step i to rank i % N" if rotate else "; root: rank 0"))
                       if dist_on else "none (1 GPU)",
                       "emulated_peers": ({"peers": emu_peers, "workgroups_per_peer": args.emulate_wgs,
                                           "channels_per_peer": CMSG, "role": args.emulate_role,
                                           "bytes_written_per_step": emu_peers * (CMSG * a_stride * 4 + CMSG * 16),
                                           "note": "sizing aid: what that many ranks' receives would write into "
                                                   "rank 0's buffers, every step, on the gather's stream"}
                                          if emu_peers else None),
                       "host_ms_per_step": {"submit": round(host_ms["process"], 3),
                                            "wait": round(host_ms.get("wait", 0.0), 3),
                                            "collect_rds": round(host_ms["collect"], 3),
                                            "note": "collect_rds includes waiting for the device (the "
                                                    "host runs ahead and blocks on the call three steps back)"}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "k_if_fir (cFineTuner + cDownsampleFilter complex)",
                         "avg_ms": round(fir_ms, 4), "launches_averaged": calls,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "valu_tflops_nofma": round(flops_per_launch / (fir_ms * 1e-3) / 1e12, 2)},
        }
        # HBM bytes per launch from the PMC counters of the committed profile (same workload only;
        # PMC needs its own rocprofv3 passes, it cannot be collected inside this run)
        # the file is keyed by the kernel form: two tiles per workgroup (k_if_fir_mt) is what runs when
        # calls overlap beside the whole-CU serial stage, one tile (k_if_fir) otherwise
        # (above 8192 channels the batch runs as sub-batches of equal size: fmd_batch_create)
        n_sub = (C + 8191) // 8192
        c_sub = C if n_sub == 1 else min(C, ((C + n_sub - 1) // n_sub + 127) // 128 * 128)
        out["roofline"]["launches_per_call"] = n_sub
        form = "k_if_fir_mt" if (args.concurrency == 2 and 1024 <= (c_sub + 63) // 64 * 64 <= 8192
                                 and args.workload == "config4" and args.fir_reduction == 0) else "k_if_fir"
        # (two outputs per lane, k_if_fir_mt3, unless a --debug-set fir_ro says otherwise; the traffic file
        # keeps the older key)
        fir_ro = 2  # the library's default (fmd_batch.hip: dbg_fir_ro)
        for kv in (args.debug_set or []):
            if kv.split("=")[0].strip() == "fir_ro":
                fir_ro = int(kv.split("=")[1])
        label = "k_if_fir_mt3" if form == "k_if_fir_mt" and fir_ro in (2, 3) else form
        out["roofline"]["kernel"] = ("%s (cFineTuner + cDownsampleFilter complex%s)"
                                     % (label, ", ReadAsyncCB byte conversion" if u8 else ""))
        tpath = os.path.join(ROOT, "profiles", "traffic_k_if_fir.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get(form)
            if (t and args.workload == "config4" and not u8 and t.get("channels") == C
                    and t.get("samples_per_call") == N):
                out["roofline"]["traffic"] = t["bytes_per_launch"]
                out["roofline"]["traffic_source"] = t["source"]
        # what the whole path moves per call against what the algorithm needs (input once, audio once)
        algo_call = samples_per_step * ((in_bytes / cpc if shared else in_bytes)) + C * nf * 4.0
        out["algorithmic_bytes_per_call"] = int(algo_call)
        fpath = os.path.join(ROOT, "profiles", "traffic_per_call.json")
        out["fabric_bytes_per_call"] = None
        if os.path.exists(fpath) and args.workload == "config4" and not u8 and not args.debug_set:
            t = json.load(open(fpath))
            if t.get("channels") == C and t.get("samples_per_call") == N:
                out["fabric_bytes_per_call"] = t["bytes_per_call"]
                out["fabric_bytes_source"] = t["source"]
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if args.debug_set:
            out["config"]["debug_set"] = args.debug_set
        if verify is not None:
            out["verify"] = verify
        if per_rank is not None:
            out["per_rank"] = per_rank
            if gth is not None:
                seen = [p_.get("rccl_ranks_seen") for p_ in per_rank]
                out["rccl_ranks_seen"] = min(seen)  # == n_gpus when RCCL really connected every rank
                out["rccl_ranks_distinct"] = len({p_.get("rccl_rank") for p_ in per_rank})
            out["per_rank_note"] = ("ms_per_step: each rank's own clock around the timed region (value uses "
                                    "the max); gather_ms_per_step: the two gather calls of a step on the "
                                    "side stream (device events; host time of the staging copies with gloo)")
        if serial_probe is not None:
            out["serial_probe_last_8_launches"] = serial_probe
        if stage_all:
            out["stage_ms"] = {k: round(v, 4) for k, v in stage_all.items()}
            if stage_all.get("if_fir", 0) > 0:
                a_ms = stage_all["if_fir"]
                out["roofline"]["alone"] = {"avg_ms": round(a_ms, 4),
                                            "achieved": round(bytes_per_launch / (a_ms * 1e-3) / 1e9, 1),
                                            "frac": round(bytes_per_launch / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "note": "the FIR with nothing else on the chip (4 serialised steps "
                                                    "after the timed region; one tile per workgroup there, two "
                                                    "beside the serial stage); avg_ms / achieved / frac above "
                                                    "are inside the overlapped pipeline"}
        if args.workload == "config5":
            # 362 flop per input sample against 8.17 B: far on the VALU side of the ridge.  A
            # lane owns an output and adds its 4096 taps in the reference's order, no FMA: one wave per SIMD
            # (the 127 KB window fills the CU's LDS) issuing packed mul / add pairs
            r = out["roofline"]
            r["hbm"] = {"achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac"]}
            r.update(bound="valu-issue", achieved=r["valu_tflops_nofma"], peak=78.6, unit="TFLOP/s",
                     frac=round(r["valu_tflops_nofma"] / 78.6, 4))
            r["peak_note"] = "packed FP32 multiply / add without FMA: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz"
            if "alone" in r:
                r["alone"]["hbm_frac"] = r["alone"].pop("frac")
                r["alone"]["hbm_achieved_GBps"] = r["alone"].pop("achieved")
            r["issue_floor"] = ("a lone wave per SIMD issues a packed op every 8 cycles at best: 16 cycles "
                                              "per tap and output; measured 17.8 (profiles/r3_lone_wave_issue_costs.txt, "
                                              "r3_long_filter_layouts.txt); 78.6 TF = 256 CUs x 4 SIMDs x 32 flop/clk "
                                              "x 2.4 GHz needs two waves per SIMD, which the window does not leave room for")
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(if_filter_order=order)
        # RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would
        # otherwise come out at exit, behind the JSON line: flush it first so that the line is the last one
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

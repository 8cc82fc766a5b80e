"""What the host process has to provide: nothing -- the library reads no environment
variable, and with HIP's default number of hardware queues the probe at fmd_batch_create finds every internal
stream a queue of its own.  A host that LOWERS `GPU_MAX_HW_QUEUES` is told so: fmd_batch_streams_sharing_queue()
counts the streams that share, and fmd_batch_create leaves a sentence in fmd_last_error() while returning FMD_OK."""
import os
import subprocess
import sys

import pytest

from __graft_entry__ import ROOT

pytestmark = pytest.mark.gpu

PROBE = r"""
import sys
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
b = pkg.Batch(pkg.make_params(2.4e6, -0.36e6, 48000.0, 15000.0, 11), 1024, record_callbacks=False)
print("SHARING", b.streams_sharing_queue())
print("MESSAGE", pkg.lib().fmd_last_error().decode())
b.close()
"""


def _probe(queues):
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    if queues is not None:
        env["GPU_MAX_HW_QUEUES"] = str(queues)
    out = subprocess.run([sys.executable, "-c", PROBE % ROOT], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict(l.split(" ", 1) for l in out.stdout.splitlines() if l.startswith(("SHARING", "MESSAGE")))
    return int(lines["SHARING"]), lines.get("MESSAGE", "")


def test_default_hardware_queues_are_enough():
    sharing, _ = _probe(None)
    assert sharing == 0


def test_a_host_that_lowers_the_queue_count_is_told():
    sharing, message = _probe(1)
    assert sharing > 0
    assert "share a hardware queue" in message and "GPU_MAX_HW_QUEUES" in message


def test_callers_stream_shares_no_queue_with_the_internal_streams():
    """fmd_batch_debug_stream_conflicts: the create-time probe against the CALLER'S stream -- null stream or a created
    one, nothing of the batch's five streams queues behind it or in front of it with HIP's default queue count (so what
    a created caller's stream costs, docs/MEASUREMENTS.md round 6, is not a shared hardware queue) -- not even with ONE
    queue per priority: the internal streams are created at the high and the low priority, whose queues are not the
    normal priority's."""
    code = r"""
import sys
sys.path.insert(0, %r)
import torch
from __graft_entry__ import load_package
pkg = load_package()
b = pkg.Batch(pkg.make_params(2.4e6, -0.36e6, 48000.0, 15000.0, 11), 1024, record_callbacks=False)
s = torch.cuda.Stream()
print("NULL", b.debug_stream_conflicts(None))
print("OWN", b.debug_stream_conflicts(s.cuda_stream))
b.close()
""" % ROOT

    def run(queues):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if queues is not None:
            env["GPU_MAX_HW_QUEUES"] = str(queues)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
        assert out.returncode == 0, out.stderr[-2000:]
        return {l.split()[0]: int(l.split()[1]) for l in out.stdout.splitlines() if l.startswith(("NULL", "OWN"))}

    assert run(None) == {"NULL": 0, "OWN": 0}
    assert run(1) == {"NULL": 0, "OWN": 0}

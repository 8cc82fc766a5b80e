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

"""Impaired stations (detuned, weak, noisy, over-deviated, silent, hot, marginal pilot / RDS) and
ragged call sizes through the HIP batch path and the CPU oracle, every block bit for bit
(tools/soak.py; longer runs by hand: `python tools/soak.py 32 120 1`)."""
import os
import subprocess
import sys

import pytest

from __graft_entry__ import ROOT

pytestmark = pytest.mark.gpu


def test_soak_impaired_stations():
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "16", "36", "3"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 mismatches" in out.stdout


def test_fuzz_random_geometries():
    """Random IF rates, downsample factors, IF filter orders, tuner table sizes and call sizes
    (tools/fuzz_geometries.py; 600 geometries run by hand in round 2 without a mismatch)."""
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_geometries.py"), "25", "11"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 with a mismatch" in out.stdout

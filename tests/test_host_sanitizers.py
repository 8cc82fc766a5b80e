"""The product's host-only code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available
on the pool; the device code's guard is the bit-exact comparison of everything it writes): the filter design
(csrc/fmd_design.hpp) over a sweep of constructor parameters -- IF rate, downsample 1-56, PCM rate, bandwidth, IF
filter order 1-8192, tuner table 1-1000; refusals are exceptions, never undefined behaviour -- and the UECP group
decoder (csrc/fmd_groups.hpp) on arbitrary groups with resets and station changes.  tests/cpp/host_sanitize.cpp;
`host_sanitize full` is ten times the sweep (run once: 153 216 designs, 2 000 000 groups, no report)."""
import os
import shutil
import subprocess

import pytest

from __graft_entry__ import ROOT


def test_design_and_group_decoder_under_asan_and_ubsan(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-ffp-contract=off",
                            os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "asan" in build.stderr.lower():
        pytest.skip("the sanitizer runtimes are not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "designs made" in run.stdout and not run.stderr.strip(), run.stderr

"""The history stays source-only: built objects (librdhip.so, the tuning harness binaries under csrc/tune/) travel to the GPU box through gpurun's
snapshot but are never tracked -- an `add -A` swept some in twice."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_built_binaries_are_tracked():
    try:
        files = subprocess.run(["git", "ls-files", "-z"], cwd=ROOT, capture_output=True, check=True, timeout=60).stdout.split(b"\0")
    except (OSError, subprocess.SubprocessError):
        pytest.skip("not a git checkout")
    bad = []
    for f in files:
        p = os.path.join(ROOT, f.decode())
        if f and os.path.isfile(p):
            with open(p, "rb") as fh:
                if fh.read(4) == b"\x7fELF":
                    bad.append(f.decode())
    assert not bad, f"tracked ELF files: {bad}"

"""Runs the C++ host-mirror test binary (erased-cells_amd/host/test_host_mirror.cpp):
the reference's unit tests restated against the compiled C++ mirror of its API."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "erased-cells_amd", "host")
BIN = os.path.join(HOST, "test_host_mirror")


def _build():
    subprocess.check_call(["make", "-C", HOST, "-s"])


def test_host_mirror_lattice_and_scalars_no_gpu():
    _build()
    r = subprocess.run([BIN, "--host-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks passed" in r.stdout


@pytest.mark.gpu
def test_host_mirror_reference_tests_on_gpu():
    if not os.path.exists(BIN):
        _build()
    env = dict(os.environ, TEST_DATA_DIR=os.path.join(ROOT, "tests", "golden"))  # the reference's fixtures
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks passed" in r.stdout and "host-only" not in r.stdout
    assert "incl. GDAL fixture tests" in r.stdout

"""polar_mult (bpmf_amd/csrc/philox.h): the device's replacement for `sqrt(-2 * log(r2) / r2)` of the polar method
(libstdc++ bits/random.tcc, normal_distribution::operator(); the reference reaches it through c++/mvnormal.cpp:41-43).
The same source compiled for the host (the hardware reciprocal / reciprocal-square-root seeds replaced by 23-bit
stand-ins) is compared with the long-double evaluation of the expression over the polar method's own distribution of
r2, a log-uniform sweep down to 2^-104 (the smallest r2 two canonical doubles can produce) and the edge values: at
most 2 ulp from the exact value (the libm expression itself is up to ~1.2 ulp away).  The device stream is checked
against the oracle in tests/test_gpu_parity.py::test_device_normal_stream."""
import os
import subprocess
import tempfile

from tests.conftest import ROOT


def test_polar_mult_is_within_two_ulp_of_the_exact_factor():
    src = os.path.join(ROOT, "tools", "probes", "polar_mult_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "pmc")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", src, "-o", exe])
        r = subprocess.run([exe, "1000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "polar_mult(1) = -0" in r.stdout or "polar_mult(1) = 0" in r.stdout

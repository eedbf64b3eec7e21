"""The host-side sanitizer run SURVEY.md §5 asks for: the oracle built with -fsanitize=address,undefined (oracle/Makefile
`asan`) replays the golden trajectories in a child process; any heap error or undefined behaviour aborts the child.
(GPU sanitizers are not available on this pool; the device code is covered by bit-exact parity against this oracle.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_golden_suite_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan):
        pytest.skip("libasan not installed")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1", LD_PRELOAD=libasan,
               QECMC_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_golden_surf.py"),
                        os.path.join(ROOT, "tests", "test_oracle_golden_alpha.py"), os.path.join(ROOT, "tests", "test_oracle_golden_planar.py"),
                        os.path.join(ROOT, "tests", "test_oracle_golden_ptdc.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr

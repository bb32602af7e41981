"""The production host side is Fortran: galahad_amd/fortran/gsls_iface.f90 (ISO_C_BINDING) +
gsls_kat.f90, the 5x5 known-answer system of src/sls/slss.f90 driven from Fortran.
CPU: the module compiles with amdflang, links against libgsls.so, analyse reproduces the reference's
statistics (15 entries, 55 flops) and the numeric phase fails loudly (-51) without a device.
GPU: the driver prints the reference's golden line ' Solution is 1.00 2.00 3.00 4.00 5.00'
(src/sls/slsds.output)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "galahad_amd", "fortran")
EXE = os.path.join(FDIR, "gsls_kat")


def _build():
    if not os.path.exists(EXE):
        if shutil.which("amdflang") is None:
            pytest.skip("amdflang not available and gsls_kat not prebuilt")
        subprocess.run(["bash", os.path.join(FDIR, "build.sh")], check=True)


def test_fortran_binding_analyse_and_loud_failure(have_gpu):
    _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert "analyse ok, entries_in_factors = 15 flops = 55" in p.stdout
    if not have_gpu:
        assert p.returncode == 2 and "flag -51" in p.stdout


@pytest.mark.gpu
def test_fortran_binding_known_answer():
    _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " Solution is  1.00  2.00  3.00  4.00  5.00" in p.stdout
    assert "negative eigenvalues = 2" in p.stdout and "PASS" in p.stdout

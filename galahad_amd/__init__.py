"""galahad_amd: MI355X-native sparse symmetric factorize+solve backend for GALAHAD's SLS/SBLS path.

The product is the HIP library behind include/gsls.h (libgsls.so) and the Fortran binding in
galahad_amd/fortran/; `galahad_amd.sls` mirrors the SLS façade for the parity tests.
Importing this package requires the built library -- there is no fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError loudly when libgsls.so is absent)
from .sls import SLS, SMT, Control, InformSLS  # noqa: F401

__all__ = ["SLS", "SMT", "Control", "InformSLS"]

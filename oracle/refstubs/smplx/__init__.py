"""Import stub standing in for the absent third-party `smplx` package when the UNMODIFIED reference is executed in the
build container (oracle/ref_harness.py).  All arithmetic lives in oracle/smplx_lbs.py."""
from oracle.smplx_lbs import SMPLLayer as SMPL  # noqa: F401
from . import lbs  # noqa: F401

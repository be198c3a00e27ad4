"""Stand-in for mpi4py used ONLY by oracle/capture_multirank.py (build container): several
"ranks" of the reference run as threads of one Python process.  TEST INFRASTRUCTURE."""
__version__ = '3.1.0'
from . import MPI  # noqa: F401,E402

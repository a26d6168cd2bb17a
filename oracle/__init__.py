"""CPU oracle for the slam3d registration hot path — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
See oracle/s3d_oracle.h for the parity status (unpinned: PCL is not available).
"""
from .oracle import *  # noqa: F401,F403

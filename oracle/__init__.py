"""CPU parity oracle -- TEST INFRASTRUCTURE ONLY (see gq_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from .gq_oracle import *  # noqa: F401,F403

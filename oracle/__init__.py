"""CPU checker for the BLASTed hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See blasted_oracle.h for the parity statement ("pinned to tolerance by the reference's own
known-answer fixtures"; the reference itself cannot be built in this image).
"""
from .oracle import *  # noqa: F401,F403

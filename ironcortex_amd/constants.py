"""Numerics contract shared by every kernel and host module (reference constants.py:1-4)."""
EPS_DIV = 1e-8    # guards divisions and the decay floor
EPS_LOG = 1e-12   # guards logarithms
MAX_EXP = 40.0    # clamp applied before exp()
MAX_NOISE = 1.0   # upper bound on the learned noise levels

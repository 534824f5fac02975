import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_hip_steer import run_steps
for steer in (True, False):
    out, c, loop, n = run_steps("factorized", steer, n_steps=2)
    print("steer", steer, c, n, out[0][0]["total"], out[1][0]["total"])

"""Where do the device-steered and host-steered train_steps differ?  (debug aid for tests/test_hip_steer.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_hip_steer import run_steps
name = sys.argv[1] if len(sys.argv) > 1 else "default"
a, ca, la, na = run_steps(name, True, n_steps=1)
b, cb, lb, nb = run_steps(name, False, n_steps=1)
print("counters", ca, cb, na, nb)
(ma, ga, sa, wa, va), (mb, gb, sb, wb, vb) = a[0], b[0]
for k in ma:
    if ma[k] != mb[k]:
        print("metric", k, ma[k], mb[k])
for k in ga:
    if (ga[k] is None) != (gb[k] is None):
        print("None-ness", k, ga[k] is None, gb[k] is None)
    elif ga[k] is not None and not torch.equal(ga[k], gb[k]):
        print("grad", k, float((ga[k] - gb[k]).abs().max()), float(gb[k].abs().max()))
for k in sa:
    if not torch.equal(sa[k], sb[k]):
        print("state", k, float((sa[k].float() - sb[k].float()).abs().max()))
print("slots equal", torch.equal(wa, wb), "value_bias", va, vb)
for i, (x, y) in enumerate(zip(la[:3], lb[:3])):
    print("loop out", i, torch.equal(x, y))
for i, (x, y) in enumerate(zip(la[:3], lb[:3])):
    if not torch.equal(x, y):
        dd = (x.float() - y.float()).abs()
        print("loop out", i, "max diff", float(dd.max()), "per sample", [float(v) for v in dd.reshape(dd.shape[0], -1).max(1).values])
for j, (ta, tb) in enumerate(zip(la[3], lb[3])):
    print("traces", j, len(ta), len(tb), [bool(torch.equal(x, y)) for x, y in zip(ta, tb)])

"""Observed-error log of the parity tests: every comparison made through
`check()` records the error it actually saw next to the tolerance it was held to, as JSON lines in
gpurun_out/parity_errors_<tag>.jsonl (merged back from the GPU box; the summaries quoted in
DESIGN.md §4 are copied to profiles/)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_worst = {}


def rel_err(a, b) -> float:
    """max|a-b| relative to the reference tensor's largest magnitude (scalars: to |b|)."""
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b), dtype=torch.float64)
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def check(tag: str, what: str, a, b, tol: float, floor: float = 0.0):
    """assert rel_err(a, b) <= tol (differences below `floor` absolute always pass) and remember
    the worst error seen per (tag, quantity class)."""
    e = rel_err(a, b)
    bb = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b), dtype=torch.float64)
    aa = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a), dtype=torch.float64)
    absd = float((aa - bb).abs().max()) if aa.numel() else 0.0
    cls = what.split(":")[0]
    w = _worst.setdefault(tag, {})
    if absd > floor and e > w.get(cls, (0.0, ""))[0]:
        w[cls] = (e, what)
    elif absd <= floor and floor > 0:  # comparisons settled by the absolute floor: remember the largest
        k = cls + "(abs, under floor %g)" % floor
        if absd >= w.get(k, (0.0, ""))[0]:
            w[k] = (absd, what)
    assert e <= tol or absd <= floor, f"{tag} {what}: rel err {e:.3e} (abs {absd:.3e}) > {tol:.1e}"
    return e


def note(tag: str, cls: str, err: float, what: str = ""):
    """Remember an error measured by the caller's own metric."""
    w = _worst.setdefault(tag, {})
    if err >= w.get(cls, (-1.0, ""))[0]:
        w[cls] = (err, what)


def flush(tag: str):
    w = _worst.pop(tag, None)
    if not w:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rec = {"case": tag, "worst_rel_err": {k: {"err": v[0], "at": v[1]} for k, v in sorted(w.items())}}
    with open(os.path.join(out, "parity_errors.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("\n[parity]", json.dumps(rec))

"""Host logic of the LoopEngine that needs no GPU: argument-struct images."""
import ctypes

import numpy as np


def test_struct_image_linear_fields():
    from ironcortex_amd.engine import StructImage

    class Inner(ctypes.Structure):
        _fields_ = [("n", ctypes.c_int32), ("pad", ctypes.c_int32), ("p", ctypes.c_void_p), ("q", ctypes.c_void_p)]

    class Outer(ctypes.Structure):
        _fields_ = [("f", Inner), ("g", ctypes.c_void_p), ("x", ctypes.c_float), ("k", ctypes.c_int32)]

    o = Outer()
    o.f.n, o.f.p, o.f.q, o.g, o.x, o.k = 7, 1000, 5, 2000, 1.5, 3
    img = StructImage(o)
    img.linear(("f", "p"), 1000, 64)   # advances 64 bytes per step
    img.linear(("g",), 9000, -8)       # any stride, negative too
    rows = img.images(range(4))
    assert rows.shape == (4, ctypes.sizeof(Outer) // 8)
    for t in range(4):
        o.f.p, o.g = 0, 0
        img.load(rows[t])
        assert (o.f.n, o.f.p, o.f.q, o.g, o.k) == (7, 1000 + 64 * t, 5, 9000 - 8 * t, 3)
        assert o.x == 1.5
    # fields set after the snapshot are overwritten by load(): patch them afterwards, per step
    o.k = 99
    img.load(rows[1])
    assert o.k == 3
    assert isinstance(rows, np.ndarray) and rows.dtype == np.int64


def test_fast_attrs_keep_module_semantics():
    """FastAttrs only shortcuts nn.Module.__setattr__ for plain bookkeeping names and already
    registered buffers: buffers stay buffers (state_dict, .to()), parameters and sub-modules still
    go through nn.Module, and a tensor assigned to a fast buffer name replaces the buffer."""
    import torch
    import torch.nn as nn
    from ironcortex_amd.utils import FastAttrs

    class M(FastAttrs, nn.Module):
        _plain_attrs = frozenset({"note"})
        _fast_buffers = frozenset({"buf"})

        def __init__(self):
            super().__init__()
            self.note = 1
            self.register_buffer("buf", torch.zeros(3))
            self.lin = nn.Linear(2, 2)
            self.w = nn.Parameter(torch.ones(2))

    m = M()
    assert "note" in m.__dict__ and m.note == 1
    m.note = torch.ones(1)  # stays a plain attribute, whatever the value
    assert "note" in m.__dict__ and "note" not in dict(m.named_buffers())
    m.buf = torch.arange(3.0)
    assert "buf" not in m.__dict__ and torch.equal(dict(m.named_buffers())["buf"], torch.arange(3.0))
    assert torch.equal(m.state_dict()["buf"], torch.arange(3.0))
    assert set(dict(m.named_parameters())) == {"w", "lin.weight", "lin.bias"} and isinstance(m.lin, nn.Linear)
    m.buf = None  # not a tensor: nn.Module's own path (keeps the registration)
    assert "buf" in m._buffers and m._buffers["buf"] is None
    m2 = M().double()
    assert m2.buf.dtype == torch.float64

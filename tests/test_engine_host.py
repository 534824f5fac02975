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

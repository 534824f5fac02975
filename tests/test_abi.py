"""CPU checks of the C-ABI boundary: the library loads, exports every symbol declared in
include/ironcortex_hip.h, and the ctypes struct layouts equal the C compiler's."""
import ctypes
import os
import subprocess
import sys

import pytest

from ironcortex_amd import hip


def test_library_exports_every_declared_symbol():
    lib = hip.lib()
    assert len(hip.FUNCTIONS) >= 19
    for fn in hip.FUNCTIONS:
        assert hasattr(lib, fn), fn
    assert lib.ic_version() >= 1
    assert lib.ic_bwd_nslab(256) == 8 and lib.ic_bwd_nslab(512) == 64


def test_struct_layouts_match_c(tmp_path):
    names = sorted(hip.CLASSES)
    src = tmp_path / "sz.c"
    body = "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names)
    src.write_text('#include <stdio.h>\n#include "ironcortex_hip.h"\nint main(){' + body + "return 0;}")
    exe = tmp_path / "sz"
    inc = os.path.dirname(hip.HEADER_PATH)
    subprocess.check_call(["gcc", "-I", inc, str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:
        assert int(out[n]) == ctypes.sizeof(hip.CLASSES[n]), n


def test_errors_are_reported_not_thrown():
    lib = hip.lib()
    rc = lib.ic_kwta_fwd(None, None, None, 1, 256, 4, 0, ctypes.c_float(1.0), None)
    assert rc == hip.ENUMS["IC_EINVAL"]
    assert b"NULL" in lib.ic_last_error()
    a = hip.new("ic_region_args", S=1, R=1, d=6)
    assert lib.ic_region_phase_fwd(ctypes.byref(a), None) == hip.ENUMS["IC_EINVAL"]
    assert b"d=6" in lib.ic_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from ironcortex_amd.utils import KWTA
    with pytest.raises(RuntimeError):
        KWTA(torch.randn(2, 8), 2)

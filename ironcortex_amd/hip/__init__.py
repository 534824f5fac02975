"""ctypes binding of libironcortex_hip.so (the C ABI in include/ironcortex_hip.h).

The struct layouts are parsed from the header at import time, so the Python side can
never drift from the C declarations.  There is no CPU fallback: if the shared library is
missing or a call returns an error code, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REPO_DIR = os.path.dirname(_PKG_DIR)
# IC_LIB_PATH: another build of the same library (A/B timing of a kernel change on one box)
LIB_PATH = os.environ.get("IC_LIB_PATH") or os.path.join(_PKG_DIR, "libironcortex_hip.so")
HEADER_PATH = os.path.join(_REPO_DIR, "include", "ironcortex_hip.h")

_SCALARS = {
    "int32_t": ctypes.c_int32,
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "uint8_t": ctypes.c_uint8,
    "int64_t": ctypes.c_int64,
}


def _strip_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def parse_header(path: str = HEADER_PATH):
    """Return ({struct name: [(field, ctype)]}, {enum name: value}, [function names])."""
    text = _strip_comments(open(path).read())
    enums: Dict[str, int] = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [t.strip() for t in item.split("=", 1)]
                nxt = int(eval(val, {}, dict(enums)))  # literals / shifts of literals only
            else:
                name = item
            enums[name] = nxt
            nxt += 1
    structs: Dict[str, List[Tuple[str, object]]] = {}
    classes: Dict[str, type] = {}
    for name, body in re.findall(r"typedef struct (\w+)\s*\{(.*?)\}\s*\w+;", text, flags=re.S):
        fields: List[Tuple[str, object]] = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            decl = decl.replace("const ", "")
            m = re.match(r"(\w+)\s*(.*)", decl)
            base, rest = m.group(1), m.group(2)
            all_ptr = "*" in decl  # the header never mixes pointers and values in one declaration
            for var in rest.split(","):
                var = var.replace("*", "").strip()
                arr = re.match(r"(\w+)\[(\d+)\]$", var)  # fixed-size array member
                if arr:
                    var = arr.group(1)
                if all_ptr:
                    ct = ctypes.c_void_p
                elif base in _SCALARS:
                    ct = _SCALARS[base]
                elif base in classes:
                    ct = classes[base]
                else:
                    raise ValueError(f"cannot parse field '{decl}' in struct {name}")
                fields.append((var, ct * int(arr.group(2)) if arr else ct))
        structs[name] = fields
        classes[name] = type(name, (ctypes.Structure,), {"_fields_": fields})
    funcs = re.findall(r"^\s*(?:const char\*|int)\s+(ic_\w+)\s*\(", text, flags=re.M)
    return structs, classes, enums, funcs


STRUCTS, CLASSES, ENUMS, FUNCTIONS = parse_header()
globals().update(ENUMS)

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C ironcortex_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback for the hot path"
            )
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ic_last_error.restype = ctypes.c_char_p
        _lib.ic_prof_kernel_name.restype = ctypes.c_char_p
        for fn in FUNCTIONS:
            if fn not in ("ic_last_error", "ic_prof_kernel_name"):
                getattr(_lib, fn).restype = ctypes.c_int
        vp = ctypes.c_void_p  # raw device addresses are passed as Python ints: declare them 64-bit
        _lib.ic_branch_score.argtypes = [vp, vp, vp, vp]
        _lib.ic_gemv_chain.argtypes = [vp, ctypes.c_int32, vp]
        _lib.ic_gemv_multi.argtypes = [vp, ctypes.c_int32, vp, vp, vp]
        _lib.ic_front_fwd.argtypes = [vp, vp, vp]
        _lib.ic_loop_step_fwd.argtypes = [vp, vp, vp, vp]
        _lib.ic_heads_gemv_fwd.argtypes = [vp, vp, vp, vp]
        _lib.ic_verify_score.argtypes = [vp, ctypes.c_int32, vp, vp, vp]
        # entry points that take bare pointers and scalars: declared in full, so that a Python int handed over as a
        # device address can never be truncated to a C int (a 32-bit "pointer" reaching a kernel is a GPU memory fault)
        i32, f32 = ctypes.c_int32, ctypes.c_float
        _lib.ic_kwta_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp]
        _lib.ic_kwta_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]
        _lib.ic_rmsnorm_fwd.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp]
        _lib.ic_rmsnorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
        _lib.ic_goodness_fwd.argtypes = [vp, vp, i32, i32, vp]
        _lib.ic_goodness_bwd.argtypes = [vp, vp, vp, i32, i32, vp]
        _lib.ic_hebbian_rank1.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        for fn_ in ("ic_region_phase_fwd", "ic_region_phase_bwd", "ic_region_wgrad", "ic_router_fwd", "ic_router_bwd", "ic_router_wgrad",
                    "ic_gate_scores", "ic_gate_select", "ic_gate_fwd", "ic_heads_fwd", "ic_heads_fwd_static", "ic_steer_select",
                    "ic_loop_steer", "ic_linear_bwd", "ic_attn_fwd", "ic_attn_bwd"):
            getattr(_lib, fn_).argtypes = [vp, vp]  # (struct pointer, stream)
        _lib.ic_graph_begin.argtypes = [vp]
        _lib.ic_graph_end.argtypes = [vp, vp, vp]
        _lib.ic_graph_launch.argtypes = [vp, vp]
        _lib.ic_graph_destroy.argtypes = [vp]
    return _lib


def check(rc: int, who: str = "") -> None:
    if rc != 0:
        msg = lib().ic_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{who or 'ironcortex_hip'} failed (code {rc}): {msg}")


def new(struct_name: str, **kw):
    """Instantiate a C struct; tensors are converted to device pointers."""
    obj = CLASSES[struct_name]()
    for k, v in kw.items():
        set_field(obj, k, v)
    return obj


def ptr(t) -> int:
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("ironcortex_amd HIP ops need tensors on a ROCm device (got a CPU tensor)")
    if not t.is_contiguous():
        raise RuntimeError("ironcortex_amd HIP ops need contiguous tensors")
    return t.data_ptr()


def set_field(obj, name: str, value) -> None:
    import torch

    if isinstance(value, torch.Tensor):
        value = ptr(value)
    setattr(obj, name, value)


def stream_handle():
    import torch

    # raw handle of torch's current stream (0.1 us; torch.cuda.current_stream() costs ~3 us)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


PROF_ON = False  # (mirrors the library's flag: launch sequences are not captured in graphs while it is set)


def prof_enable(on: bool) -> None:
    global PROF_ON
    PROF_ON = bool(on)
    lib().ic_prof_enable(ctypes.c_int(1 if on else 0))


def prof_select(name: str = "", stride: int = 1) -> None:
    """Time only the named kernel ("" = all), one launch in `stride`."""
    l = lib()
    check(l.ic_prof_stride(ctypes.c_int(max(1, stride))), "ic_prof_stride")
    kid = -1
    if name:
        names = [l.ic_prof_kernel_name(ctypes.c_int(k)).decode() for k in range(l.ic_prof_kernel_count())]
        kid = names.index(name)
    check(l.ic_prof_select(ctypes.c_int(kid)), "ic_prof_select")


def prof_bytes_read() -> dict:
    """{kernel name: (launches, algorithmic bytes)} counted by the library since the previous read
    (shape-determined kernels only: GEMV chains, heads, steer)."""
    out = {}
    l = lib()
    for kid in range(l.ic_prof_kernel_count()):
        n, b = ctypes.c_int64(0), ctypes.c_double(0.0)
        check(l.ic_prof_bytes_read(ctypes.c_int(kid), ctypes.byref(n), ctypes.byref(b)), "ic_prof_bytes_read")
        if n.value:
            out[l.ic_prof_kernel_name(ctypes.c_int(kid)).decode()] = (n.value, b.value)
    return out


def prof_read() -> dict:
    """{kernel name: (launches, total_ms)} since the previous read (synchronises)."""
    out = {}
    l = lib()
    for kid in range(l.ic_prof_kernel_count()):
        n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        check(l.ic_prof_read(ctypes.c_int(kid), ctypes.byref(n), ctypes.byref(ms)), "ic_prof_read")
        if n.value:
            out[l.ic_prof_kernel_name(ctypes.c_int(kid)).decode()] = (n.value, ms.value)
    return out

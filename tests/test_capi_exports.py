"""CPU-side check of the drop-in boundary: libpphip.so builds, loads, and exports every
symbol include/pp_hip.h declares.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from pathplanning_amd import build
    path = build.build(verbose=False)
    return C.CDLL(path)


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    lib.pp_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    rc = lib.pp_ctx_create(0, None, C.byref(ctx))
    assert rc < 0
    lib.pp_last_error.restype = C.c_char_p
    assert b"no HIP device" in lib.pp_last_error() or b"hip" in lib.pp_last_error().lower()


def test_product_never_touches_the_oracle():
    """Nothing under pathplanning_amd/ may import, include or link oracle/ (test infrastructure)."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pathplanning_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                s = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle/|ppo_|libppo|oracle_lib", s):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_nonholo_dims_host_logic(lib):
    """pp_nonholo_dims is pure host arithmetic (heuristics.cpp:13-14,43-51): 1024^2 map -> 103 x 103 x 73."""
    from pathplanning_amd._lib import HybridParams
    import numpy as np
    lib.pp_nonholo_dims.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(HybridParams), C.c_void_p, C.c_void_p]
    p = HybridParams.default()
    dims = np.zeros(3, dtype=np.int32)
    offs = np.zeros(2)
    for half, want in ((51.2, (103, 103, 73)), (25.6, (53, 53, 73)), (10.0, (21, 21, 73)), (204.8, (411, 411, 73))):
        lo = np.array([-half, -half, -3.141592653589793])
        up = np.array([half, half, 3.141592653589793])
        rc = lib.pp_nonholo_dims(lo.ctypes.data, up.ctypes.data, C.byref(p), dims.ctypes.data, offs.ctypes.data)
        assert rc == 0
        assert tuple(dims) == want
        assert offs[0] == (want[0] // 2) * 1.0

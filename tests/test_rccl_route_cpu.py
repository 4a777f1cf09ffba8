"""The set-up checks of the RCCL reduction route that need no GPU (reduce_rccl.cpp; the route itself runs in tests/test_gpu_dropin.py)."""
import ctypes as C

import cases


def test_rccl_route_refuses_what_it_cannot_run(monkeypatch):
    """mcgpu_rccl_create is where the scan's fallback chain learns that the vendor collective cannot be used: a device listed twice
    (RCCL wants one rank per GPU) and the test hook of the chain both answer -1 with a reason, before the library is even opened."""
    eng = cases.pkg.engine
    lib = eng.load_library()
    lib.mcgpu_rccl_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    lib.mcgpu_last_error.restype = C.c_char_p
    out = C.c_void_p()
    assert lib.mcgpu_rccl_create((C.c_int * 2)(0, 0), 2, C.byref(out)) == -1 and not out.value
    assert b"listed twice" in lib.mcgpu_last_error()
    monkeypatch.setenv("MCGPU_RCCL_FAIL", "1")
    assert lib.mcgpu_rccl_create((C.c_int * 2)(0, 1), 2, C.byref(out)) == -1 and not out.value
    assert b"MCGPU_RCCL_FAIL" in lib.mcgpu_last_error()
    monkeypatch.delenv("MCGPU_RCCL_FAIL")
    monkeypatch.setenv("MCGPU_RCCL_LIBRARY", "/nonexistent/librccl.so")
    assert lib.mcgpu_rccl_create((C.c_int * 2)(0, 1), 2, C.byref(out)) == -1 and not out.value
    assert b"cannot open the RCCL library" in lib.mcgpu_last_error()
    assert lib.mcgpu_rccl_create(None, 0, C.byref(out)) == -1

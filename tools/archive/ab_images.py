#!/usr/bin/env python3
"""Dev tool (GPU): dump FAST images of the bench geometry with a given engine library (A/B comparisons of kernels).
usage: ab_images.py <libmcgpu_amd.so> <out.npz> [histories] [projections...]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
eng = cases.pkg.engine
import ctypes as C
lib = C.CDLL(sys.argv[1])  # bare binding: only the entry points this tool needs (old builds lack newer symbols)
vp, cp, ci, cull = C.c_void_p, C.c_char_p, C.c_int, C.c_ulonglong
lib.mcgpu_last_error.restype = cp
lib.mcgpu_create.argtypes = [cp, ci, C.POINTER(vp)]
lib.mcgpu_destroy.argtypes = [vp]
lib.mcgpu_config_i64.argtypes = [vp, cp, C.POINTER(C.c_longlong)]
lib.mcgpu_image_words.argtypes = [vp, C.POINTER(C.c_size_t)]
lib.mcgpu_run_projection.argtypes = [vp, ci, ci, ci, cull, cull, ci, vp, C.POINTER(C.c_double), C.POINTER(cull)]
eng._lib = lib
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 3_000_000
projs = [int(a) for a in sys.argv[4:]] or [300]
out = {}
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    for p in projs:
        img, secs, done = ctx.run_projection(p, n, mode="fast", seed=42)
        out[f"p{p}"] = img
np.savez_compressed(sys.argv[2], **out)

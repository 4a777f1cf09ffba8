"""ctypes binding of the engine's C ABI (`include/mcgpu_amd.h`, built as `libmcgpu_amd.so`).

The product path: there is no CPU fallback -- if the HIP extension is missing, loading fails
loudly.  PyTorch is used only by callers that want device tensors / `torch.distributed`; this
module itself needs no torch.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

MODE_FAST, MODE_COMPAT, MODE_FAST_STATS, MODE_FAST_F64 = 0, 1, 2, 3
_MODES = {"fast": MODE_FAST, "compat": MODE_COMPAT, "stats": MODE_FAST_STATS, "fast64": MODE_FAST_F64, MODE_FAST: MODE_FAST,
          MODE_COMPAT: MODE_COMPAT, MODE_FAST_STATS: MODE_FAST_STATS, MODE_FAST_F64: MODE_FAST_F64}

LIB_PATH = Path(os.environ.get("MCGPU_AMD_LIB", Path(__file__).resolve().parent / "libmcgpu_amd.so"))  # override: A/B builds
EXE_PATH = Path(__file__).resolve().parent / "MC-GPU_v1.3.x"

# every symbol declared in include/mcgpu_amd.h
ABI_SYMBOLS = (
    "mcgpu_abi_version", "mcgpu_knob_table", "mcgpu_last_error", "mcgpu_create", "mcgpu_clone", "mcgpu_destroy", "mcgpu_config_i64", "mcgpu_config_f64",
    "mcgpu_host_table", "mcgpu_projection_file_name", "mcgpu_image_words", "mcgpu_launch_shape", "mcgpu_advance_seed",
    "mcgpu_launch_projection", "mcgpu_scheduler_stats", "mcgpu_scheduler_stats_ex", "mcgpu_last_kernel_ms", "mcgpu_clear_image", "mcgpu_run_projection",
    "mcgpu_write_projection", "mcgpu_format_projection", "mcgpu_write_formatted_projection", "mcgpu_dose_info", "mcgpu_dose_read", "mcgpu_dose_clear", "mcgpu_write_dose_report",
    "mcgpu_finalize_projection", "mcgpu_finalize_projection_host", "mcgpu_stack_create", "mcgpu_stack_append", "mcgpu_stack_write_slice", "mcgpu_stack_finish",
    "mcgpu_stack_read", "mcgpu_normalize_stack", "mcgpu_run_scan", "mcgpu_run_scan_multi", "mcgpu_set_projection_angles", "mcgpu_set_geometry_arrays",
    "mcgpu_warp_volume", "mcgpu_warp_geometry",
    "mcgpu_write_voxel_file", "mcgpu_write_voxel_binary", "mcgpu_kat_rng", "mcgpu_kat_rng_streams", "mcgpu_microbench", "mcgpu_kat_math", "mcgpu_kat_expf", "mcgpu_kat_f32", "mcgpu_kat_fast64", "mcgpu_kat_tile_records", "mcgpu_fdk_reconstruct", "mcgpu_set_fast_schedule", "mcgpu_reload_env_knobs",
    "mcgpu_exchange_shared_bytes", "mcgpu_exchange_card_bytes", "mcgpu_exchange_create", "mcgpu_exchange_card", "mcgpu_exchange_connect",
    "mcgpu_exchange_connect_local", "mcgpu_exchange_probe", "mcgpu_exchange_owner", "mcgpu_exchange_begin", "mcgpu_exchange_submit", "mcgpu_exchange_collect",
    "mcgpu_exchange_stats", "mcgpu_exchange_destroy", "mcgpu_copy_to_host",
    "mcgpu_rccl_create", "mcgpu_rccl_reduce_u64", "mcgpu_rccl_destroy",
)


class ScanOptions(C.Structure):
    """mcgpu_scan_options (include/mcgpu_amd.h)."""
    _fields_ = [("struct_size", C.c_uint), ("mode", C.c_int), ("first_projection", C.c_int), ("num_projections", C.c_int),
                ("histories_per_projection", C.c_ulonglong), ("crop_nx", C.c_int), ("write_ascii", C.c_int), ("write_stacks", C.c_int),
                ("output_folder", C.c_char_p), ("air_stack", C.c_char_p), ("air_sigma_y", C.c_double), ("air_sigma_x", C.c_double),
                ("pixel_spacing_x", C.c_double), ("pixel_spacing_y", C.c_double),
                ("shared_stacks", C.POINTER(C.c_void_p)), ("slice_of_projection", C.POINTER(C.c_int)), ("progress", C.c_int),
                ("shard", C.c_int), ("projection_stride", C.c_int), ("projection_phase", C.c_int), ("reduce", C.c_int)]


class ScanReport(C.Structure):
    """mcgpu_scan_report (include/mcgpu_amd.h)."""
    _fields_ = [("projections", C.c_int), ("histories_per_projection", C.c_ulonglong), ("seconds_total", C.c_double),
                ("seconds_kernels", C.c_double), ("seconds_after_last_kernel", C.c_double), ("zero_replacement", C.c_float * 3),
                ("seconds_writer", C.c_double), ("kernel_ms_min", C.c_double), ("kernel_ms_max", C.c_double)]


def knob_table():
    """The engine's environment knobs (csrc/knobs.cpp) as a list of dicts {name, type, scope, default, current, what}."""
    lib = load_library()
    n = lib.mcgpu_knob_table(None, 0)
    buf = C.create_string_buffer(n)
    lib.mcgpu_knob_table(buf, n)
    keys = ("name", "type", "scope", "default", "current", "what")
    return [dict(zip(keys, line.split("\t"))) for line in buf.value.decode().split("\n") if line]


class EngineError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


_lib = None


def load_library(path: Optional[os.PathLike] = None):
    """dlopen the engine and declare the prototypes.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise ImportError(f"{p} not found: build the HIP engine first (python __graft_entry__.py or make -C 4d-cbct-mc_amd/csrc)")
    lib = C.CDLL(str(p))
    missing = [s for s in ABI_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{p} lacks C-ABI symbols: {missing}")
    vp, cp, ci, cull = C.c_void_p, C.c_char_p, C.c_int, C.c_ulonglong
    lib.mcgpu_last_error.restype = cp
    lib.mcgpu_knob_table.argtypes = [C.c_char_p, C.c_size_t]
    lib.mcgpu_knob_table.restype = C.c_size_t
    lib.mcgpu_create.argtypes = [cp, ci, C.POINTER(vp)]
    lib.mcgpu_clone.argtypes = [vp, ci, C.POINTER(vp)]
    lib.mcgpu_destroy.argtypes = [vp]
    lib.mcgpu_destroy.restype = None
    lib.mcgpu_config_i64.argtypes = [vp, cp, C.POINTER(C.c_longlong)]
    lib.mcgpu_config_f64.argtypes = [vp, cp, C.POINTER(C.c_double)]
    lib.mcgpu_host_table.argtypes = [vp, cp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.mcgpu_projection_file_name.argtypes = [vp, ci, cp, C.c_size_t]
    lib.mcgpu_image_words.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.mcgpu_launch_shape.argtypes = [cull, ci, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(cull)]
    lib.mcgpu_advance_seed.argtypes = [ci, cull, ci]
    lib.mcgpu_launch_projection.argtypes = [vp, ci, ci, ci, cull, cull, ci, vp, vp]
    lib.mcgpu_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.mcgpu_scheduler_stats.argtypes = [vp, C.POINTER(cull), ci]
    lib.mcgpu_scheduler_stats_ex.argtypes = [vp, C.POINTER(cull), ci, ci]
    lib.mcgpu_clear_image.argtypes = [vp, vp, vp]
    lib.mcgpu_run_projection.argtypes = [vp, ci, ci, ci, cull, cull, ci, vp, C.POINTER(C.c_double), C.POINTER(cull)]
    lib.mcgpu_write_projection.argtypes = [vp, ci, vp, cull, C.c_double, cp]
    lib.mcgpu_format_projection.argtypes = [vp, vp, cull, ci, vp]
    lib.mcgpu_write_formatted_projection.argtypes = [vp, ci, ci, cull, C.c_double, cp]
    lib.mcgpu_dose_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(C.c_size_t)]
    lib.mcgpu_dose_read.argtypes = [vp, vp, vp]
    lib.mcgpu_dose_clear.argtypes = [vp]
    lib.mcgpu_write_dose_report.argtypes = [vp, vp, vp, cull, C.c_double, cp, C.c_size_t]
    lib.mcgpu_finalize_projection.argtypes = [vp, vp, cull, ci, vp, ci, vp]
    lib.mcgpu_finalize_projection_host.argtypes = [vp, vp, cull, ci, vp]
    lib.mcgpu_stack_create.argtypes = [cp, ci, ci, ci, C.c_double, C.c_double, C.POINTER(vp)]
    lib.mcgpu_stack_append.argtypes = [vp, vp]
    lib.mcgpu_stack_write_slice.argtypes = [vp, ci, vp]
    lib.mcgpu_set_projection_angles.argtypes = [vp, ci, C.POINTER(C.c_float)]
    lib.mcgpu_set_geometry_arrays.argtypes = [vp, C.POINTER(ci), C.POINTER(C.c_float), vp, vp]
    lib.mcgpu_warp_volume.argtypes = [vp, C.POINTER(ci), vp, vp, vp, ci, C.c_float, vp, vp]
    lib.mcgpu_warp_geometry.argtypes = [vp, vp, ci, ci, C.c_float]
    lib.mcgpu_stack_finish.argtypes = [vp, ci, C.POINTER(C.c_float)]
    lib.mcgpu_stack_read.argtypes = [cp, C.POINTER(ci), vp, C.c_size_t]
    lib.mcgpu_normalize_stack.argtypes = [cp, cp, C.c_double, C.c_double, cp, C.c_double, C.c_double]
    lib.mcgpu_run_scan.argtypes = [vp, C.POINTER(ScanOptions), C.POINTER(ScanReport)]
    lib.mcgpu_run_scan_multi.argtypes = [C.POINTER(vp), ci, C.POINTER(ScanOptions), C.POINTER(ScanReport)]
    lib.mcgpu_write_voxel_file.argtypes = [cp, C.POINTER(ci), C.POINTER(C.c_float), vp, vp, ci]
    lib.mcgpu_write_voxel_binary.argtypes = [cp, C.POINTER(ci), C.POINTER(C.c_float), vp, vp]
    lib.mcgpu_kat_rng.argtypes = [vp, ci, ci, ci, ci, ci, vp]
    lib.mcgpu_microbench.argtypes = [vp, ci, vp, ci]
    lib.mcgpu_kat_rng_streams.argtypes = [vp, ci, C.c_uint, C.c_uint, C.c_ulonglong, vp, ci, ci, vp]
    lib.mcgpu_reload_env_knobs.argtypes = [vp]
    lib.mcgpu_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.mcgpu_exchange_shared_bytes.argtypes = [ci]
    lib.mcgpu_exchange_shared_bytes.restype = C.c_size_t
    lib.mcgpu_exchange_card_bytes.argtypes = [ci]
    lib.mcgpu_exchange_card_bytes.restype = C.c_size_t
    lib.mcgpu_exchange_create.argtypes = [ci, ci, ci, C.c_size_t, ci, vp, C.POINTER(vp)]
    lib.mcgpu_exchange_card.argtypes = [vp, vp, C.c_size_t]
    lib.mcgpu_exchange_connect.argtypes = [vp, ci, vp, C.c_size_t]
    lib.mcgpu_exchange_connect_local.argtypes = [vp, vp]
    lib.mcgpu_exchange_owner.argtypes = [vp, C.c_longlong]
    lib.mcgpu_exchange_probe.argtypes = [vp]
    lib.mcgpu_exchange_begin.argtypes = [vp, C.c_longlong, vp, C.POINTER(vp)]
    lib.mcgpu_exchange_submit.argtypes = [vp, C.c_longlong, vp]
    lib.mcgpu_exchange_collect.argtypes = [vp, C.c_longlong, vp, C.POINTER(vp)]
    lib.mcgpu_exchange_stats.argtypes = [vp, C.POINTER(C.c_double)]
    lib.mcgpu_exchange_destroy.argtypes = [vp]
    lib.mcgpu_exchange_destroy.restype = None
    lib.mcgpu_kat_math.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    lib.mcgpu_kat_expf.argtypes = [vp, ci, vp, vp]
    lib.mcgpu_kat_f32.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.mcgpu_kat_fast64.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    lib.mcgpu_kat_tile_records.argtypes = [ci, vp, vp]
    if path is None:
        _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise EngineError(rc, load_library().mcgpu_last_error().decode(errors="replace"))


def launch_shape(histories: int, threads_per_block: int, histories_per_thread: int):
    """(blocks, histories_per_thread, total_histories) of the reference launch (MC-GPU_v1.3.cu:823-841)."""
    b, h, t = C.c_int(), C.c_int(), C.c_ulonglong()
    _check(load_library().mcgpu_launch_shape(int(histories), threads_per_block, histories_per_thread, C.byref(b), C.byref(h), C.byref(t)))
    return b.value, h.value, t.value


def advance_seed(batch_number: int, total_histories: int, seed: int) -> int:
    return load_library().mcgpu_advance_seed(batch_number, int(total_histories), seed)


def write_voxel_file(path, n, spacing_cm, material_zyx: np.ndarray, density_zyx: np.ndarray, gzip: bool = True):
    m = np.ascontiguousarray(material_zyx, dtype=np.uint8)
    d = np.ascontiguousarray(density_zyx, dtype=np.float32)
    assert m.size == d.size == int(n[0]) * int(n[1]) * int(n[2])
    _check(load_library().mcgpu_write_voxel_file(str(path).encode(), (C.c_int * 3)(*map(int, n)), (C.c_float * 3)(*map(float, spacing_cm)),
                                                 m.ctypes.data, d.ctypes.data, int(bool(gzip))))


def write_voxel_binary(path, n, spacing_cm, material_zyx: np.ndarray, density_zyx: np.ndarray):
    """Binary sidecar (`geometry.voxbin`) the engine prefers over the text voxel file of the same stem."""
    m = np.ascontiguousarray(material_zyx, dtype=np.uint8)
    d = np.ascontiguousarray(density_zyx, dtype=np.float32)
    assert m.size == d.size == int(n[0]) * int(n[1]) * int(n[2])
    _check(load_library().mcgpu_write_voxel_binary(str(path).encode(), (C.c_int * 3)(*map(int, n)), (C.c_float * 3)(*map(float, spacing_cm)),
                                                   m.ctypes.data, d.ctypes.data))


def kat_tile_records(indices) -> np.ndarray:
    """Tile records of `indices[n_tiles, 64]` (palette index per voxel of a 4x4x4 tile, negative = padding) as the engine builds them
    (include/mcgpu_amd.h: mcgpu_kat_tile_records): uint32[n_tiles, 4] = entries, code, mask low, mask high."""
    v = np.ascontiguousarray(indices, dtype=np.int16).reshape(-1, 64)
    out = np.zeros((v.shape[0], 4), dtype=np.uint32)
    _check(load_library().mcgpu_kat_tile_records(v.shape[0], v.ctypes.data, out.ctypes.data))
    return out


def voxel_sidecar_path(voxel_file) -> Path:
    s = str(voxel_file)
    for ext in (".gz", ".vox"):
        if s.endswith(ext):
            s = s[: -len(ext)]
    return Path(s + ".voxbin")


class StackWriter:
    """MetaImage float32 stack written plane by plane (the reference's `projections_to_itk` + `sitk.WriteImage`)."""

    def __init__(self, path, nx: int, ny: int, nslices: int, spacing=(0.776, 0.776)):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.mcgpu_stack_create(str(path).encode(), nx, ny, nslices, float(spacing[0]), float(spacing[1]), C.byref(h)))
        self.h, self.shape = h, (ny, nx)

    def append(self, plane: np.ndarray):
        a = np.ascontiguousarray(plane, dtype=np.float32)
        assert a.shape == self.shape
        _check(self.lib.mcgpu_stack_append(self.h, a.ctypes.data))

    def write_slice(self, k: int, plane: np.ndarray):
        """Random-access write (each slice once; not mixed with append): 4-D scans fill the stack grouped by respiratory state."""
        a = np.ascontiguousarray(plane, dtype=np.float32)
        assert a.shape == self.shape
        _check(self.lib.mcgpu_stack_write_slice(self.h, int(k), a.ctypes.data))

    def finish(self, replace_zeros: bool = True) -> float:
        v = C.c_float()
        h, self.h = self.h, None
        _check(self.lib.mcgpu_stack_finish(h, int(replace_zeros), C.byref(v)))
        return v.value


def stack_read(path) -> np.ndarray:
    """float32 [nslices, ny, nx] of a MetaImage stack written by this engine (or SimpleITK, uncompressed)."""
    lib = load_library()
    dims = (C.c_int * 3)()
    _check(lib.mcgpu_stack_read(str(path).encode(), dims, None, 0))
    out = np.zeros((dims[2], dims[1], dims[0]), dtype=np.float32)
    _check(lib.mcgpu_stack_read(str(path).encode(), dims, out.ctypes.data, out.size))
    return out


def normalize_stack(total_stack, air_stack, out_stack, sigma=(10.0, 10.0), spacing=(0.776, 0.776)):
    """log(gaussian_filter(air, sigma) / total) -> out_stack (cbctmc/mc/projection.py:96-115); sigma None/0 = no filter."""
    sy, sx = (0.0, 0.0) if not sigma else (float(sigma[0]), float(sigma[1]))
    _check(load_library().mcgpu_normalize_stack(str(total_stack).encode(), str(air_stack).encode(), sy, sx, str(out_stack).encode(),
                                                float(spacing[0]), float(spacing[1])))


EXCHANGE_ROOT0, EXCHANGE_ROTATE, EXCHANGE_LOCAL = 0, 1, 2


class Exchange:
    """One rank's end of the tally exchange between the GPUs of one node (mcgpu_exchange_*, include/mcgpu_amd.h): per step
    `begin` -> launch into the returned buffer -> `submit` -> `collect(previous step)`.

    `shared`: a writable buffer of `Exchange.shared_bytes(world)` zeroed bytes seen by every rank -- an `mmap` of one
    /dev/shm file between processes (`Exchange.open_shared`), a bytearray between contexts of one process."""

    def __init__(self, device: int, rank: int, world: int, words: int, shared, policy: int = EXCHANGE_ROOT0):
        self.lib = load_library()
        self._shared = shared  # keeps the mapping alive
        self._buf = (C.c_char * len(shared)).from_buffer(shared)
        if len(shared) < self.shared_bytes(world):
            raise ValueError("shared region too small")
        h = C.c_void_p()
        _check(self.lib.mcgpu_exchange_create(int(device), int(rank), int(world), int(words), int(policy), C.addressof(self._buf), C.byref(h)))
        self.h, self.rank, self.world, self.words, self.policy = h, rank, world, words, policy

    @staticmethod
    def shared_bytes(world: int) -> int:
        return int(load_library().mcgpu_exchange_shared_bytes(int(world)))

    @staticmethod
    def open_shared(path, world: int, create: bool):
        """Map the host region of an exchange between processes: rank 0 creates (and zeroes) the file BEFORE the others open it.
        The creator replaces whatever sits at `path` with a NEW inode of mode 0600 (O_EXCL | O_NOFOLLOW: a symlink planted in the
        world-writable directory is not followed, and a rank left over from a crashed run on the same port keeps its old inode
        instead of sharing counters with this run)."""
        import mmap
        n = Exchange.shared_bytes(world)
        if create:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
            os.ftruncate(fd, n)  # zero-filled
        else:
            fd = os.open(path, os.O_RDWR | os.O_NOFOLLOW)
        try:
            return mmap.mmap(fd, n)
        finally:
            os.close(fd)

    def card(self) -> bytes:
        n = int(self.lib.mcgpu_exchange_card_bytes(self.world))
        buf = C.create_string_buffer(n)
        _check(self.lib.mcgpu_exchange_card(self.h, buf, n))
        return buf.raw

    def connect(self, peer: int, card: bytes):
        _check(self.lib.mcgpu_exchange_connect(self.h, int(peer), card, len(card)))

    def connect_local(self, other: "Exchange"):
        _check(self.lib.mcgpu_exchange_connect_local(self.h, other.h))

    def probe(self):
        """One small copy-engine transfer to every connected peer, waited for (before the first step)."""
        _check(self.lib.mcgpu_exchange_probe(self.h))

    def owner(self, step: int) -> int:
        return int(self.lib.mcgpu_exchange_owner(self.h, int(step)))

    def begin(self, step: int, stream: int = 0) -> int:
        p = C.c_void_p()
        _check(self.lib.mcgpu_exchange_begin(self.h, int(step), C.c_void_p(stream), C.byref(p)))
        return p.value

    def submit(self, step: int, stream: int = 0):
        _check(self.lib.mcgpu_exchange_submit(self.h, int(step), C.c_void_p(stream)))

    def collect(self, step: int, stream: int = 0) -> Optional[int]:
        """Device pointer of the complete tally of `step` on its owner (valid until begin(step + 2)), None elsewhere."""
        p = C.c_void_p()
        _check(self.lib.mcgpu_exchange_collect(self.h, int(step), C.c_void_p(stream), C.byref(p)))
        return p.value

    def stats(self) -> dict:
        out = (C.c_double * 6)()
        _check(self.lib.mcgpu_exchange_stats(self.h, out))
        return {"last_push_ms": out[0], "last_add_ms": out[1], "pushes": int(out[2]), "collects": int(out[3]), "host_wait_s": out[4], "bytes_per_push": int(out[5])}

    def close(self):
        if getattr(self, "h", None):
            self.lib.mcgpu_exchange_destroy(self.h)
            self.h = None
            del self._buf

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One loaded simulation (input file + tables), optionally resident on one GPU."""

    def __init__(self, input_path, device: int = 0, _clone_of: "Context" = None):
        self.lib = load_library()
        h = C.c_void_p()
        if _clone_of is not None:
            _check(self.lib.mcgpu_clone(_clone_of.h, int(device), C.byref(h)))
        else:
            _check(self.lib.mcgpu_create(str(input_path).encode(), int(device), C.byref(h)))
        self.h = h
        self.device = device
        self.input_path = str(input_path)

    def clone(self, device: int = 0) -> "Context":
        """The same simulation on another device without parsing the input files again (mcgpu_clone)."""
        return Context(self.input_path, device, _clone_of=self)

    # -- lifecycle
    def close(self):
        if getattr(self, "h", None):
            self.lib.mcgpu_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- queries
    def geti(self, key: str) -> int:
        v = C.c_longlong()
        _check(self.lib.mcgpu_config_i64(self.h, key.encode(), C.byref(v)))
        return v.value

    def getf(self, key: str) -> float:
        v = C.c_double()
        _check(self.lib.mcgpu_config_f64(self.h, key.encode(), C.byref(v)))
        return v.value

    def host_table(self, name: str, dtype=np.uint8) -> np.ndarray:
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib.mcgpu_host_table(self.h, name.encode(), C.byref(p), C.byref(n)))
        buf = (C.c_char * n.value).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).copy()

    @property
    def num_projections(self) -> int:
        return self.geti("num_projections")

    @property
    def image_words(self) -> int:
        n = C.c_size_t()
        _check(self.lib.mcgpu_image_words(self.h, C.byref(n)))
        return n.value

    @property
    def detector_shape(self):
        return self.geti("num_pixels_z"), self.geti("num_pixels_x")

    def projection_file_name(self, p: int) -> str:
        buf = C.create_string_buffer(1024)
        _check(self.lib.mcgpu_projection_file_name(self.h, p, buf, 1024))
        return buf.value.decode()

    # -- running
    def launch(self, p: int, image_dev_ptr: int, count: int, mode="fast", seed: Optional[int] = None, first: int = 0,
               hpt: Optional[int] = None, stream: int = 0):
        """Asynchronous launch adding into a caller-owned device buffer (e.g. a torch int64 tensor's data_ptr())."""
        seed = self.geti("seed") if seed is None else seed
        hpt = self.geti("histories_per_thread") if hpt is None else hpt
        _check(self.lib.mcgpu_launch_projection(self.h, p, _MODES[mode], int(seed), int(first), int(count), int(hpt),
                                                C.c_void_p(image_dev_ptr), C.c_void_p(stream)))

    def clear(self, image_dev_ptr: int, stream: int = 0):
        _check(self.lib.mcgpu_clear_image(self.h, C.c_void_p(image_dev_ptr), C.c_void_p(stream)))

    def download_image(self, image_dev_ptr: int, stream: int = 0) -> np.ndarray:
        """uint64[4, Nz, Nx] copy of a device tally (waits for `stream`)."""
        img = np.zeros(self.image_words, dtype=np.uint64)
        _check(self.lib.mcgpu_copy_to_host(self.h, C.c_void_p(image_dev_ptr), img.ctypes.data, img.nbytes, C.c_void_p(stream)))
        nz, nx = self.detector_shape
        return img.reshape(4, nz, nx)

    def reload_env_knobs(self):
        """Read the MCGPU_* tuning knobs of the environment again (they are otherwise read once, at creation)."""
        _check(self.lib.mcgpu_reload_env_knobs(self.h))

    def scheduler_stats(self, reset: bool = True) -> dict:
        """Counters of "stats"-mode launches (diagnostic build): mean flying lanes per wave iteration etc."""
        out = (C.c_ulonglong * 32)()
        _check(self.lib.mcgpu_scheduler_stats_ex(self.h, out, 32, int(reset)))
        names = ("iterations", "flying_lanes", "compton_rounds", "compton_lanes", "rayleigh_rounds", "rayleigh_lanes", "new_rounds", "new_lanes",
                 "scheduling_points", "take_rounds", "take_lanes", "drain_points", "cycles_compton", "cycles_rayleigh", "cycles_new", "cycles_flight",
                 "compton_angle_lanes", "compton_shell_lanes", "compton_done_lanes", "pool_flyable", "pool_wants_new", "pool_compton", "slots_traded", "lanes_taking_a_step",
                 "iter_with_voxel_load", "voxel_load_lanes", "iter_with_sigma_load", "sigma_load_lanes",
                 "cycles_flight_to_voxel", "cycles_flight_resolve", "cycles_settle", "cycles_sched_point")
        return dict(zip(names, [int(v) for v in out]))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(self.lib.mcgpu_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def run_projection(self, p: int, count: int, mode="fast", seed: Optional[int] = None, first: int = 0, hpt: Optional[int] = None):
        """Synchronous: returns (image uint64[4, Nz, Nx], kernel_seconds, histories_done)."""
        seed = self.geti("seed") if seed is None else seed
        hpt = self.geti("histories_per_thread") if hpt is None else hpt
        img = np.zeros(self.image_words, dtype=np.uint64)
        secs, done = C.c_double(), C.c_ulonglong()
        _check(self.lib.mcgpu_run_projection(self.h, p, _MODES[mode], int(seed), int(first), int(count), int(hpt), img.ctypes.data,
                                             C.byref(secs), C.byref(done)))
        nz, nx = self.detector_shape
        return img.reshape(4, nz, nx), secs.value, done.value

    def reference_shape(self, histories: Optional[int] = None):
        """(batches, hpt, total_histories) the reference would launch for `histories` (COMPAT mode units)."""
        histories = self.geti("total_histories") if histories is None else histories
        tpb, hpt = self.geti("threads_per_block"), self.geti("histories_per_thread")
        blocks, hpt, total = launch_shape(histories, tpb, hpt)
        return blocks * tpb, hpt, total

    def write_projection(self, p: int, image: np.ndarray, total_histories: int, seconds: float = 0.0, file_name: Optional[str] = None):
        img = np.ascontiguousarray(image, dtype=np.uint64).reshape(-1)
        assert img.size == self.image_words
        _check(self.lib.mcgpu_write_projection(self.h, p, img.ctypes.data, int(total_histories), float(seconds),
                                               file_name.encode() if file_name else None))
        return file_name or self.projection_file_name(p)

    def write_projection_device(self, p: int, image_dev_ptr: int, total_histories: int, seconds: float = 0.0, file_name: Optional[str] = None,
                                slot: int = 0, stream: int = 0):
        """The same file with the data lines formatted on the device from a device tally (mcgpu_format_projection +
        mcgpu_write_formatted_projection); synchronises `stream` in between."""
        _check(self.lib.mcgpu_format_projection(self.h, C.c_void_p(image_dev_ptr), int(total_histories), int(slot), C.c_void_p(stream)))
        import torch
        torch.cuda.synchronize()
        _check(self.lib.mcgpu_write_formatted_projection(self.h, p, int(slot), int(total_histories), float(seconds), file_name.encode() if file_name else None))
        return file_name or self.projection_file_name(p)

    # -- post-processing (cbctmc/mc/projection.py) and the whole-scan pipeline
    def finalize_host(self, image: np.ndarray, total_histories: int, crop_nx: int = 0) -> np.ndarray:
        """float32 [3, Nz, crop] = (total, unscattered, scattered), z flipped -- what the reference's Python gets from the ASCII file."""
        img = np.ascontiguousarray(image, dtype=np.uint64).reshape(-1)
        nz, nx = self.detector_shape
        cx = crop_nx if 0 < crop_nx < nx else nx
        out = np.zeros((3, nz, cx), dtype=np.float32)
        _check(self.lib.mcgpu_finalize_projection_host(self.h, img.ctypes.data, int(total_histories), int(crop_nx), out.ctypes.data))
        return out

    def finalize_device(self, image_dev_ptr: int, total_histories: int, planes_dev_ptr: int, crop_nx: int = 0, clear: bool = False, stream: int = 0):
        _check(self.lib.mcgpu_finalize_projection(self.h, C.c_void_p(image_dev_ptr), int(total_histories), int(crop_nx), C.c_void_p(planes_dev_ptr),
                                                  int(clear), C.c_void_p(stream)))

    # -- 4-D: several (geometry, projection angles) jobs on one resident context (cbctmc/mc/simulation.py:527-710)
    def set_projection_angles(self, angles_deg):
        """Explicit projection angles [deg]; pose 0 stays the input file's (pass the first angle twice like the reference
        and start the scan at first_projection=1)."""
        a = np.ascontiguousarray(angles_deg, dtype=np.float32)
        _check(self.lib.mcgpu_set_projection_angles(self.h, int(a.size), a.ctypes.data_as(C.POINTER(C.c_float))))

    def set_geometry_arrays(self, n, spacing_cm, material_zyx: np.ndarray, density_zyx: np.ndarray):
        """Replace the voxel volume from arrays ([z][y][x]); material tables are rebuilt and everything is uploaded again."""
        m = np.ascontiguousarray(material_zyx, dtype=np.uint8)
        d = np.ascontiguousarray(density_zyx, dtype=np.float32)
        assert m.size == d.size == int(n[0]) * int(n[1]) * int(n[2])
        _check(self.lib.mcgpu_set_geometry_arrays(self.h, (C.c_int * 3)(*map(int, n)), (C.c_float * 3)(*map(float, spacing_cm)),
                                                  m.ctypes.data, d.ctypes.data))

    def set_geometry(self, geometry):
        """An `MCGeometry` of the Python mirror, in the orientation its voxel file would have (geo.py:589-599)."""
        mats, dens, spacing_cm = geometry.mcgpu_arrays()
        nx, ny, nz = mats.shape
        self.set_geometry_arrays((nx, ny, nz), spacing_cm, np.transpose(mats, (2, 1, 0)), np.transpose(dens, (2, 1, 0)))

    def warp_geometry(self, displacement: np.ndarray, frame: str = "geometry", default_material: int = 1, default_density: float = 0.0013):
        """The context's geometry := warp(base geometry, displacement) entirely on the device (mcgpu_warp_geometry).
        frame "geometry": displacement [3, gx, gy, gz] in the frame of the MCGeometry arrays (what the reference's
        correspondence model predicts; the rot90 of the voxel file is handled on the device); frame "engine": [3, nz, ny, nx].
        Raises EngineError(-5) when the volume is not a palette volume: fall back to warp_volume + set_geometry."""
        u = np.ascontiguousarray(displacement, dtype=np.float32)
        nx, ny, nz = self.geti("num_voxels_x"), self.geti("num_voxels_y"), self.geti("num_voxels_z")
        want = (3, ny, nx, nz) if frame == "geometry" else (3, nz, ny, nx)
        if u.shape != want:
            raise ValueError(f"displacement of shape {u.shape}, expected {want} for frame '{frame}'")
        _check(self.lib.mcgpu_warp_geometry(self.h, u.ctypes.data, 1 if frame == "geometry" else 0, int(default_material), float(default_density)))

    def warp_volume(self, material_zyx: np.ndarray, density_zyx: np.ndarray, displacement: np.ndarray, default_material: int, default_density: float):
        """Nearest-neighbour warp on the GPU: out[x] = in[rint(x + u(x))]; displacement [3, nz, ny, nx] (x, y, z components, voxels)."""
        m = np.ascontiguousarray(material_zyx, dtype=np.uint8)
        d = np.ascontiguousarray(density_zyx, dtype=np.float32)
        u = np.ascontiguousarray(displacement, dtype=np.float32)
        nz, ny, nx = m.shape
        assert d.shape == m.shape and u.shape == (3, nz, ny, nx)
        mo, do = np.empty_like(m), np.empty_like(d)
        _check(self.lib.mcgpu_warp_volume(self.h, (C.c_int * 3)(nx, ny, nz), m.ctypes.data, d.ctypes.data, u.ctypes.data, int(default_material),
                                          float(default_density), mo.ctypes.data, do.ctypes.data))
        return mo, do

    def run_scan(self, mode="fast", first_projection=0, num_projections=0, histories=0, crop_nx=0, write_ascii=False, write_stacks=True,
                 output_folder=None, air_stack=None, air_sigma=(10.0, 10.0), pixel_spacing=(0.0, 0.0), shared_stacks=None,
                 slice_of_projection=None, peers=(), shard="histories", projection_stride=0, projection_phase=0, reduce="auto") -> dict:
        """The whole projection loop as a device/host pipeline (mcgpu_run_scan); returns the timing report.
        `shared_stacks` = three open StackWriters (total, unscattered, scattered) filled by slice index (4-D scans).
        `peers` + shard="histories": the reference's split (tallies summed through the exchange); shard="projections": every
        context simulates whole projections, nothing crosses between devices (SURVEY 8e fallback).  reduce="rccl": the tallies of the
        peers are summed with one ncclReduce per projection instead of the exchange (then projection sharding if RCCL cannot be set up)."""
        o = ScanOptions()
        o.struct_size = C.sizeof(ScanOptions)
        o.shard = {"histories": 0, "projections": 1}[shard]
        o.reduce = {"auto": 0, "rccl": 1}[reduce]
        o.projection_stride, o.projection_phase = int(projection_stride), int(projection_phase)
        if shared_stacks is not None:
            self._keep = ((C.c_void_p * 3)(*[s.h for s in shared_stacks]), (C.c_int * len(slice_of_projection))(*map(int, slice_of_projection)))
            o.shared_stacks, o.slice_of_projection = self._keep[0], self._keep[1]
        o.mode, o.first_projection, o.num_projections = _MODES[mode], int(first_projection), int(num_projections)
        o.histories_per_projection, o.crop_nx = int(histories), int(crop_nx)
        o.write_ascii, o.write_stacks = int(bool(write_ascii)), int(bool(write_stacks))
        o.output_folder = str(output_folder).encode() if output_folder else None
        o.air_stack = str(air_stack).encode() if air_stack else None
        o.air_sigma_y, o.air_sigma_x = (float(air_sigma[0]), float(air_sigma[1])) if air_sigma else (0.0, 0.0)
        o.pixel_spacing_x, o.pixel_spacing_y = float(pixel_spacing[0]), float(pixel_spacing[1])
        r = ScanReport()
        if peers:  # contexts of the same input on other devices: histories sharded, tallies reduced on this context's device
            hs = (C.c_void_p * (1 + len(peers)))(self.h, *[p.h for p in peers])
            _check(self.lib.mcgpu_run_scan_multi(hs, 1 + len(peers), C.byref(o), C.byref(r)))
        else:
            _check(self.lib.mcgpu_run_scan(self.h, C.byref(o), C.byref(r)))
        return {"projections": r.projections, "histories_per_projection": r.histories_per_projection, "seconds_total": r.seconds_total,
                "seconds_kernels": r.seconds_kernels, "seconds_after_last_kernel": r.seconds_after_last_kernel,
                "seconds_writer": r.seconds_writer, "zero_replacement": [float(v) for v in r.zero_replacement],
                "kernel_ms_min": r.kernel_ms_min, "kernel_ms_max": r.kernel_ms_max}

    # -- dose tallies (SECTION DOSE DEPOSITION of the input file)
    def dose_info(self):
        """(flags, roi6, roi_shape_zyx): flags bit 0 = material tally, bit 1 = voxel tally."""
        flags, roi, n = C.c_int(), (C.c_int * 6)(), C.c_size_t()
        _check(self.lib.mcgpu_dose_info(self.h, C.byref(flags), roi, C.byref(n)))
        r = list(roi)
        shape = (r[5] - r[4] + 1, r[3] - r[2] + 1, r[1] - r[0] + 1) if flags.value & 2 else (0, 0, 0)
        return flags.value, r, shape

    def dose_read(self):
        """(voxels uint64[Dz, Dy, Dx, 2] or None, materials uint64[25, 2] or None) accumulated on the device so far."""
        flags, _, shape = self.dose_info()
        vox = np.zeros(shape + (2,), dtype=np.uint64) if flags & 2 else None
        mat = np.zeros((25, 2), dtype=np.uint64) if flags & 1 else None
        _check(self.lib.mcgpu_dose_read(self.h, vox.ctypes.data if vox is not None else None, mat.ctypes.data if mat is not None else None))
        return vox, mat

    def dose_clear(self):
        _check(self.lib.mcgpu_dose_clear(self.h))

    def write_dose_report(self, voxels, materials, histories_per_projection: int, seconds: float = 0.0) -> str:
        """Writes the voxel dose files (when `voxels` is given) and returns the text of both reports."""
        v = np.ascontiguousarray(voxels, dtype=np.uint64) if voxels is not None else None
        m = np.ascontiguousarray(materials, dtype=np.uint64) if materials is not None else None
        log = C.create_string_buffer(1 << 16)
        _check(self.lib.mcgpu_write_dose_report(self.h, v.ctypes.data if v is not None else None, m.ctypes.data if m is not None else None,
                                                int(histories_per_projection), float(seconds), log, len(log)))
        return log.value.decode(errors="replace")

    def run_all(self, mode="fast", write_projections=True, histories: Optional[int] = None):
        """The projection loop of main() (MC-GPU_v1.3.cu:667-1056) on this context's GPU."""
        out = []
        seed = self.geti("seed")
        histories = self.geti("total_histories") if histories is None else histories
        for p in range(self.num_projections):
            if _MODES[mode] == MODE_COMPAT:
                batches, hpt, total = self.reference_shape(histories)
                img, secs, done = self.run_projection(p, batches, mode, seed=seed, hpt=hpt)
                seed = advance_seed(1, total, seed)
            else:
                img, secs, done = self.run_projection(p, histories, mode, seed=seed)
            name = self.write_projection(p, img, done, secs) if write_projections else None
            out.append((name, img, secs, done))
        return out

    def microbench(self, kind: str):
        """Hardware ceilings measured on this context's device (include/mcgpu_amd.h: mcgpu_microbench): kind "valu_issue" ->
        wave-instructions per ns and SIMD {64 lanes, lanes 0-31, 32 lanes spread}; "atomic_rate" -> scattered 64-bit adds per second."""
        out = (C.c_double * 3)()
        _check(self.lib.mcgpu_microbench(self.h, {"valu_issue": 0, "atomic_rate": 1}[kind], out, 3))
        return [float(v) for v in out] if kind == "valu_issue" else float(out[0])

    # -- known-answer hooks
    def kat_rng(self, mode, seed: int, batch: int, hpt: int, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.float32)
        _check(self.lib.mcgpu_kat_rng(self.h, _MODES[mode], seed, batch, hpt, n, out.ctypes.data))
        return out

    def kat_rng_streams(self, seed: int, projection: int, n_draws: int, first_id: int = 0, n_ids: int = 0, ids=None, generator: int = 0) -> np.ndarray:
        """uint32[n_ids, n_draws]: raw outputs of the FAST per-history streams (generator 1: the Philox-per-draw yardstick)."""
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint64)
            n_ids = ids.size
        out = np.zeros((n_ids, n_draws), dtype=np.uint32)
        _check(self.lib.mcgpu_kat_rng_streams(self.h, generator, seed, projection, first_id, ids.ctypes.data if ids is not None else None,
                                              n_ids, n_draws, out.ctypes.data))
        return out

    def kat_math(self, x: Sequence[float]):
        x = np.ascontiguousarray(x, dtype=np.float64)
        outs = [np.zeros_like(x) for _ in range(4)]
        _check(self.lib.mcgpu_kat_math(self.h, x.size, x.ctypes.data, *[o.ctypes.data for o in outs]))
        return outs

    def kat_fast64(self, u, a, b, c, directions):
        """float64[n, 8] of mcgpu_kat_fast64: sin, cos, 1/sqrt(a), sqrt(a/b), cdt1, rotated direction (3)."""
        u = np.ascontiguousarray(u, dtype=np.uint32)
        a, b, c = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, b, c))
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert u.size == a.size == b.size == c.size == d.shape[0]
        out = np.zeros((u.size, 8), dtype=np.float64)
        _check(self.lib.mcgpu_kat_fast64(self.h, u.size, u.ctypes.data, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, out.ctypes.data))
        return out

    def kat_f32(self, op: int, a, b=None, c=None):
        """Float operations of the COMPAT kernel (include/mcgpu_amd.h: mcgpu_kat_f32)."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
        out = np.ascontiguousarray(np.zeros_like(a) if c is None else c, dtype=np.float32).copy()
        _check(self.lib.mcgpu_kat_f32(self.h, int(op), a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def kat_expf(self, x: Sequence[float]):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros_like(x)
        _check(self.lib.mcgpu_kat_expf(self.h, x.size, x.ctypes.data, out.ctypes.data))
        return out


def create(input_path, device: int = 0) -> Context:
    return Context(input_path, device)

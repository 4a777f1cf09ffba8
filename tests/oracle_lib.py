"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and, when present, of the reference
harness (oracle/_ref/libmcgpu_ref.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
ORACLE_SO = ROOT / "oracle" / "liboracle.so"
REF_SO = ROOT / "oracle" / "_ref" / "libmcgpu_ref.so"
REF_EXE = ROOT / "oracle" / "_ref" / "MC-GPU_v1.3_CPU.x"

MATH_LIBM, MATH_PORTABLE = 0, 1
MAXMAT, MAXSHELLS, NPRAY, MAXRAYBINS, MAXEBINS = 25, 40, 128, 25005, 256

SOURCE_DT = np.dtype([("position", "<f4", 3), ("direction", "<f4", 3), ("rot_fan", "<f4", 9), ("cos_theta_low", "<f4"),
                      ("phi_low", "<f4"), ("D_cos_theta", "<f4"), ("D_phi", "<f4"), ("max_height_at_y1cm", "<f4")])
DETECTOR_DT = np.dtype([("sdd", "<f4"), ("lateral_displacement", "<f4"), ("corner_min_rotated_to_Y", "<f4", 3),
                        ("center", "<f4", 3), ("rot_inv", "<f4", 9), ("width_X", "<f4"), ("height_Z", "<f4"),
                        ("inv_pixel_size_X", "<f4"), ("inv_pixel_size_Z", "<f4"), ("num_pixels", "<i4", 2),
                        ("total_num_pixels", "<i4"), ("rotation_flag", "<i4")])
assert SOURCE_DT.itemsize == 80 and DETECTOR_DT.itemsize == 100


class OracleTables(C.Structure):
    _fields_ = [
        ("voxel_mat_dens", C.c_void_p), ("num_voxels", C.c_int * 3), ("inv_voxel_size", C.c_float * 3),
        ("size_bbox", C.c_float * 3), ("num_values", C.c_int), ("e0", C.c_float), ("ide", C.c_float),
        ("mfp_woodcock", C.c_void_p), ("mfp_a", C.c_void_p), ("mfp_b", C.c_void_p),
        ("xco", C.c_void_p), ("pco", C.c_void_p), ("aco", C.c_void_p), ("bco", C.c_void_p), ("pmax", C.c_void_p),
        ("itlco", C.c_void_p), ("ituco", C.c_void_p),
        ("fco", C.c_void_p), ("uico", C.c_void_p), ("fj0", C.c_void_p), ("noscco", C.c_void_p),
        ("num_bins_espc", C.c_int), ("espc", C.c_void_p), ("espc_cutoff", C.c_void_p), ("espc_alias", C.c_void_p),
        ("source_data", C.c_void_p), ("detector_data", C.c_void_p),
        ("dose_roi", C.c_int * 6), ("voxels_edep", C.c_void_p), ("materials_dose", C.c_void_p), ("image_w2", C.c_void_p),
    ]


class OracleCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("histories", "steps", "voxel_reads", "mfp_reads", "woodcock_reads", "compton",
                                          "rayleigh", "photo", "rng", "tally_calls", "tally_hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def ensure_oracle_built():
    if not ORACLE_SO.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True)
    return ORACLE_SO


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        try:  # oracle/Makefile builds with -mfma (the explicit fma() calls of gl_expf)
            if "fma" not in next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split():
                raise RuntimeError("oracle/liboracle.so needs a CPU with FMA3 (oracle/Makefile: -mfma)")
        except (OSError, StopIteration):
            pass
        lib = C.CDLL(str(ensure_oracle_built()))
        lib.oracle_track.argtypes = [C.POINTER(OracleTables), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int, C.c_int, C.POINTER(OracleCounters)]
        lib.oracle_track.restype = C.c_int
        lib.oracle_ranecu.restype = C.c_float
        lib.oracle_ranecu_double.restype = C.c_double
        lib.oracle_update_seed.argtypes = [C.c_int, C.c_ulonglong, C.c_int]
        lib.oracle_pm_log.restype = C.c_double
        lib.oracle_pm_log.argtypes = [C.c_double]
        lib.oracle_pm_exp.restype = C.c_double
        lib.oracle_pm_exp.argtypes = [C.c_double]
        lib.oracle_compton_s.restype = C.c_float
        lib.oracle_compton_s.argtypes = [C.POINTER(OracleTables), C.c_float, C.c_float, C.c_int, C.c_int]
        lib.oracle_gl_expf.restype = C.c_float
        lib.oracle_gl_expf.argtypes = [C.c_float]
        lib.oracle_pm_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.oracle_rotate.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        lib.oracle_gcoa.argtypes = [C.POINTER(OracleTables), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.oracle_graa.argtypes = [C.POINTER(OracleTables), C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.oracle_source.argtypes = [C.POINTER(OracleTables), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int]
        _oracle = lib
    return _oracle


class TableSet:
    """Host tables in the reference layouts as numpy arrays + the ctypes view passed to the oracle."""

    FIELDS = ("voxel_mat_dens", "mfp_woodcock", "mfp_a", "mfp_b", "xco", "pco", "aco", "bco", "pmax", "itlco", "ituco",
              "fco", "uico", "fj0", "noscco", "espc", "espc_cutoff", "espc_alias", "source_data", "detector_data")

    def __init__(self, arrays: dict, num_voxels, inv_voxel_size, size_bbox, num_values, e0, ide, num_bins_espc):
        self.a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        self.num_voxels = tuple(int(x) for x in num_voxels)
        self.inv_voxel_size = tuple(np.float32(x) for x in inv_voxel_size)
        self.size_bbox = tuple(np.float32(x) for x in size_bbox)
        self.num_values, self.e0, self.ide, self.num_bins_espc = int(num_values), np.float32(e0), np.float32(ide), int(num_bins_espc)
        t = OracleTables()
        for k in self.FIELDS:
            setattr(t, k, self.a[k].ctypes.data)
        t.num_voxels = (C.c_int * 3)(*self.num_voxels)
        t.inv_voxel_size = (C.c_float * 3)(*self.inv_voxel_size)
        t.size_bbox = (C.c_float * 3)(*self.size_bbox)
        t.num_values, t.e0, t.ide, t.num_bins_espc = self.num_values, float(self.e0), float(self.ide), self.num_bins_espc
        t.dose_roi = (C.c_int * 6)(32500, -32500, 32500, -32500, 32500, -32500)
        t.voxels_edep, t.materials_dose, t.image_w2 = None, None, None
        self.ct = t
        self.dose_voxels = self.dose_materials = None

    def enable_dose(self, roi6=None, materials=True):
        """Attach zeroed dose tallies (0-based inclusive ROI, or None for no voxel tally); returns (voxels, materials)."""
        if roi6 is not None:
            r = [int(x) for x in roi6]
            shape = (r[5] - r[4] + 1, r[3] - r[2] + 1, r[1] - r[0] + 1, 2)
            self.dose_voxels = np.zeros(shape, dtype=np.uint64)
            self.ct.dose_roi = (C.c_int * 6)(*r)
            self.ct.voxels_edep = self.dose_voxels.ctypes.data
        if materials:
            self.dose_materials = np.zeros((MAXMAT, 2), dtype=np.uint64)
            self.ct.materials_dose = self.dose_materials.ctypes.data
        return self.dose_voxels, self.dose_materials

    @property
    def detector(self):
        return self.a["detector_data"].view(DETECTOR_DT)

    @property
    def source(self):
        return self.a["source_data"].view(SOURCE_DT)

    def image_size(self):
        return 4 * int(self.detector[0]["total_num_pixels"])

    def track(self, num_p, seed, batch0, nbatches, hpt, math_mode=MATH_LIBM, n_threads=1, image=None, counters=None, w2=None):
        """`w2` (uint64 array like the image): also accumulate the squared tally weights, units (1024 x 0.01 eV)^2."""
        if image is None:
            image = np.zeros(self.image_size(), dtype=np.uint64)
        cnt = counters if counters is not None else OracleCounters()
        self.ct.image_w2 = w2.ctypes.data if w2 is not None else None
        try:
            oracle().oracle_track(C.byref(self.ct), num_p, seed, batch0, nbatches, hpt, image.ctypes.data, math_mode, n_threads,
                                  C.byref(cnt))
        finally:
            self.ct.image_w2 = None
        return image, cnt

    def track_with_variance(self, num_p, seed, batch0, nbatches, hpt, math_mode=MATH_LIBM, n_threads=1):
        """(image, sum of squared weights in the image's units squared as float64, counters)."""
        w2 = np.zeros(self.image_size(), dtype=np.uint64)
        image, cnt = self.track(num_p, seed, batch0, nbatches, hpt, math_mode, n_threads, w2=w2)
        return image, w2.astype(np.float64) * (1024.0 ** 2), cnt


class Reference:
    """The real reference engine (CPU build) behind oracle/ref_harness.c.  One input file per process state."""

    def __init__(self):
        if not REF_SO.exists():
            raise FileNotFoundError(str(REF_SO))
        lib = C.CDLL(str(REF_SO))
        lib.ref_get.restype = C.c_void_p
        lib.ref_get.argtypes = [C.c_char_p, C.POINTER(C.c_long)]
        lib.ref_ranecu.restype = C.c_float
        lib.ref_ranecu_double.restype = C.c_double
        lib.ref_update_seed.argtypes = [C.c_int, C.c_ulonglong, C.c_int]
        lib.ref_report.argtypes = [C.c_char_p, C.c_int, C.c_ulonglong, C.c_double]
        lib.ref_rotate_double.argtypes = [C.c_void_p, C.c_double, C.c_double]
        lib.ref_gcoa.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.ref_graa.argtypes = [C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.ref_source.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_report_dose.argtypes = [C.c_char_p, C.c_ulonglong, C.c_double]
        self.lib = lib
        self.loaded = None

    def load(self, input_path):
        self.lib.ref_load(str(input_path).encode())
        self.loaded = str(input_path)
        sc = (C.c_double * 16)()
        self.lib.ref_get_scalars(sc)
        names = ("total_histories", "seed", "gpu_id", "threads_per_block", "histories_per_thread", "num_projections", "D_angle",
                 "angularROI_0", "angularROI_1", "initial_angle", "SRotAxisD", "vertical_translation", "flag_material_dose",
                 "enable_specific_angles", "mean_energy_spectrum")
        self.scalars = dict(zip(names, list(sc)))
        return self

    def get(self, name, dtype=np.uint8, copy=True):
        n = C.c_long()
        p = self.lib.ref_get(name.encode(), C.byref(n))
        if not p or n.value < 0:
            raise KeyError(name)
        buf = (C.c_char * n.value).from_address(p)
        arr = np.frombuffer(buf, dtype=dtype)
        return arr.copy() if copy else arr

    def tables(self) -> TableSet:
        nproj = int(self.scalars["num_projections"])
        vd = self.get("voxel_data")
        num_voxels = vd[:12].view("<i4")
        inv_vs = vd[12:24].view("<f4")
        bbox = vd[24:36].view("<f4")
        mt = self.get("mfp_table_data")
        ray = self.get("rayleigh")
        o = 0
        r = {}
        for k in ("xco", "pco", "aco", "bco"):
            r[k] = ray[o:o + 4 * NPRAY * MAXMAT].view("<f4"); o += 4 * NPRAY * MAXMAT
        r["pmax"] = ray[o:o + 4 * MAXRAYBINS * MAXMAT].view("<f4"); o += 4 * MAXRAYBINS * MAXMAT
        r["itlco"] = ray[o:o + NPRAY * MAXMAT]; o += NPRAY * MAXMAT
        r["ituco"] = ray[o:o + NPRAY * MAXMAT]
        com = self.get("compton")
        n = 4 * MAXMAT * MAXSHELLS
        r["fco"], r["uico"], r["fj0"] = com[0:n].view("<f4"), com[n:2 * n].view("<f4"), com[2 * n:3 * n].view("<f4")
        r["noscco"] = com[3 * n:3 * n + 4 * MAXMAT].view("<i4")
        se = self.get("source_energy")
        nb = int(se[:4].view("<i4")[0])
        r["espc"] = se[4:4 + 4 * MAXEBINS].view("<f4")
        r["espc_cutoff"] = se[4 + 4 * MAXEBINS:4 + 8 * MAXEBINS].view("<f4")
        r["espc_alias"] = se[4 + 8 * MAXEBINS:4 + 10 * MAXEBINS].view("<i2")
        r["voxel_mat_dens"] = self.get("voxel_mat_dens", "<f4")
        r["mfp_woodcock"] = self.get("mfp_woodcock", "<f4")
        r["mfp_a"] = self.get("mfp_a", "<f4")
        r["mfp_b"] = self.get("mfp_b", "<f4")
        r["source_data"] = self.get("source_data")[:80 * nproj]
        r["detector_data"] = self.get("detector_data")[:100 * nproj]
        return TableSet(r, num_voxels, inv_vs, bbox, int(mt[:4].view("<i4")[0]), mt[4:8].view("<f4")[0], mt[8:12].view("<f4")[0], nb)

    def track(self, num_p, seed, batch0, nbatches, hpt, clear=True):
        if clear:
            self.lib.ref_clear_image()
        self.lib.ref_track(num_p, seed, batch0, nbatches, hpt)
        return self.get("image", "<u8")

    def dose(self):
        """(roi6, voxels uint64[n, 2] or None, materials uint64[25, 2]) tallied since the last clear."""
        roi = self.get("dose_roi", "<i2").astype(int).tolist()
        vox = self.get("voxels_edep", "<u8").reshape(-1, 2) if roi[1] > -1 else None
        return roi, vox, self.get("materials_dose", "<u8").reshape(MAXMAT, 2)

    def report_dose(self, out_file, total_histories, seconds=0.0):
        """report_voxels_dose (when the ROI is enabled) + report_materials_dose of the reference; stdout not captured."""
        return self.lib.ref_report_dose(str(out_file).encode(), int(total_histories), float(seconds))


def reference_available():
    return REF_SO.exists()

#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, built from /root/reference).

Test infrastructure; runs only in the development container (the reference does not exist on the
GPU box).  Usage:  make -C oracle ref && python oracle/gen_golden.py [case ...]   (default: every case + the KATs)

Fixtures written (all inputs come from tests/cases.py, so the tests can rebuild the same cases):
  golden/rng_kat.npz        RANECU known answers: init_PRNG seeds, float/double draw sequences, abMODm,
                            update_seed_PRNG
  golden/physics_kat.npz    GCOa / GRAa / rotate_double / source() outputs of the reference for seeded calls
                            (catphan64 tables)
  golden/case_<name>.npz    per case: scalars, source/detector structs, spectrum alias tables, SHA-256 digests
                            and sampled rows of the big tables, sparse integer tallies of the reference CPU
                            loop (ref_track) per projection, the same for the oracle's PORTABLE math mode
                            (machine-independent: this is what the GPU compat kernel must reproduce),
                            output file names, and the data lines of one ASCII projection file (tiny detector)
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))

import cases  # noqa: E402
import oracle_lib as ol  # noqa: E402

GOLD = ROOT / "tests" / "golden"
NBATCH = {"air": 200, "water": 200, "catphan64": 400, "catphan64_ct": 100, "slab_angles": 100, "catphan64_dose": 150, "graded_u16": 100, "graded_raw": 100,
          "cirs76": 100, "thorax64": 100, "tissue22": 100, "thorax128_bone": 100}
HPT = 150


class capture:
    """Redirect the process's stdout (C stdio included) into a file."""

    def __init__(self, path):
        self.path = str(path)

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.fd = os.open(self.path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.dup2(self.fd, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.fd)
        os.close(self.saved)


class quiet:
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.null)
        os.close(self.saved)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def sparse(img: np.ndarray):
    idx = np.flatnonzero(img).astype(np.uint32)
    return idx, img[idx].astype(np.uint64)


def rng_kat(ref):
    out = {}
    combos = [(0, 150, 42), (1, 150, 42), (2, 150, 42), (5208 * 128 - 1, 150, 42), (65000 * 128 - 1, 1431, 42), (7, 33, 123456789), (0, 1, 1)]
    seeds, draws_f, draws_d = [], [], []
    for batch, hpt, seed in combos:
        s = (C.c_int * 2)()
        ref.lib.ref_init_prng(batch, hpt, seed, s)
        seeds.append([batch, hpt, seed, s[0], s[1]])
        draws_f.append([ref.lib.ref_ranecu(s) for _ in range(1000)])
        draws_d.append([ref.lib.ref_ranecu_double(s) for _ in range(200)])
    out["init"] = np.array(seeds, dtype=np.int64)
    out["draws_f32"] = np.array(draws_f, dtype=np.float32)
    out["draws_f64"] = np.array(draws_d, dtype=np.float64)
    ab = [(2147483563, 40014, 42), (2147483399, 40692, 42), (2147483563, 2147483562, 2147483562), (2147483399, 1234567, 7654321), (2147483563, 32768, 99), (2147483563, 32769, 99)]
    out["abmodm"] = np.array([[m, a, s, ref.lib.ref_abmodm(m, a, s)] for m, a, s in ab], dtype=np.int64)
    us = [(1, 100012800, 42), (1, 11905920000, 42), (8, 1488240000, 42), (3, 150, 2147483562), (0, 5, 77)]
    out["update_seed"] = np.array([[b, h, s, ref.lib.ref_update_seed(b, h, s)] for b, h, s in us], dtype=np.int64)
    np.savez_compressed(GOLD / "rng_kat.npz", **out)
    print("rng_kat: init_PRNG(0,150,42) ->", seeds[0][3:], " update_seed(1,100012800,42) ->", out["update_seed"][0, 3])


def physics_kat(ref, T):
    """Seeded calls into the reference's GCOa/GRAa/rotate_double/source on the catphan64 tables."""
    rng = np.random.default_rng(20241220)
    used = np.flatnonzero(T.a["noscco"])
    gco, gra, rot, src = [], [], [], []
    for _ in range(400):
        mat = int(rng.choice(used))
        e = np.float32(rng.uniform(6000, 125000))
        s = (C.c_int * 2)(int(rng.integers(1, 2147483562)), int(rng.integers(1, 2147483398)))
        s0 = (s[0], s[1])
        ef, ct = C.c_float(e), C.c_double()
        ref.lib.ref_gcoa(C.byref(ef), C.byref(ct), mat, s)
        gco.append([mat, float(e), s0[0], s0[1], ef.value, ct.value, s[0], s[1]])
    for _ in range(400):
        mat = int(rng.choice(used))
        e = np.float32(rng.uniform(6000, 124990))
        idx = int((e - T.e0) * T.ide)
        s = (C.c_int * 2)(int(rng.integers(1, 2147483562)), int(rng.integers(1, 2147483398)))
        s0 = (s[0], s[1])
        ct = C.c_double()
        ref.lib.ref_graa(C.c_float(e), C.byref(ct), mat, idx, s)
        gra.append([mat, float(e), idx, s0[0], s0[1], ct.value, s[0], s[1]])
    for _ in range(400):
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        if rng.uniform() < 0.05:
            d = np.array([0, 0, 1 if rng.uniform() < 0.5 else -1], dtype=np.float32)
        costh, phi = float(rng.uniform(-1, 1)), float(rng.uniform(0, 2 * np.pi))
        buf = (C.c_float * 3)(*d)
        ref.lib.ref_rotate_double(buf, costh, phi)
        rot.append([d[0], d[1], d[2], costh, phi, buf[0], buf[1], buf[2]])
    for _ in range(400):
        s = (C.c_int * 2)(int(rng.integers(1, 2147483562)), int(rng.integers(1, 2147483398)))
        s0 = (s[0], s[1])
        pos, dr, en, av = (C.c_float * 3)(), (C.c_float * 3)(), C.c_float(), C.c_int()
        ref.lib.ref_source(0, s, pos, dr, C.byref(en), C.byref(av))
        src.append([s0[0], s0[1], pos[0], pos[1], pos[2], dr[0], dr[1], dr[2], en.value, av.value, s[0], s[1]])
    np.savez_compressed(GOLD / "physics_kat.npz", gcoa=np.array(gco, dtype=np.float64), graa=np.array(gra, dtype=np.float64),
                        rotate=np.array(rot, dtype=np.float64), source=np.array(src, dtype=np.float64))
    print("physics_kat: 4 x 400 seeded calls")


def case_fixture(ref, name, workdir):
    inp = cases.build_case(name, workdir / name)
    with quiet():
        ref.load(inp)
    T = ref.tables()
    nproj = int(ref.scalars["num_projections"])
    out = {"scalars_names": np.array(list(ref.scalars.keys())), "scalars": np.array(list(ref.scalars.values()), dtype=np.float64),
           "source_data": T.a["source_data"].copy(), "detector_data": T.a["detector_data"].copy(),
           "espc": T.a["espc"][: T.num_bins_espc + 1].copy(), "espc_cutoff": T.a["espc_cutoff"][: T.num_bins_espc].copy(),
           "espc_alias": T.a["espc_alias"][: T.num_bins_espc].copy(), "num_bins_espc": T.num_bins_espc,
           "num_voxels": np.array(T.num_voxels), "inv_voxel_size": np.array(T.inv_voxel_size, dtype=np.float32),
           "size_bbox": np.array(T.size_bbox, dtype=np.float32), "e0_ide": np.array([T.e0, T.ide], dtype=np.float32),
           "num_values": T.num_values, "noscco": T.a["noscco"].copy(), "density_max": ref.get("density_max", "<f4"),
           "density_nominal": ref.get("density_nominal", "<f4")}
    used = np.flatnonzero(T.a["noscco"])
    nv = T.num_values
    A = T.a["mfp_a"].reshape(nv, 25, 3)[:, used]
    B = T.a["mfp_b"].reshape(nv, 25, 3)[:, used]
    W = T.a["mfp_woodcock"].reshape(nv, 2)[: nv - 1]  # the reference leaves the last entry uninitialised
    rows = np.array([0, 1, 2, 1000, 12000, 23998, 23999, 24000])
    rows = rows[rows < nv]
    digests = {
        "voxel_mat_dens": sha(T.a["voxel_mat_dens"]), "mfp_a_used": sha(A), "mfp_b_used": sha(B), "woodcock_but_last": sha(W),
        "pmax_used": sha(T.a["pmax"].reshape(-1, 25)[:nv, used]),
        "rayleigh_used": sha(np.stack([T.a[k].reshape(25, 128)[used] for k in ("xco", "pco", "aco", "bco")])),
        "itl_itu_used": sha(np.stack([T.a[k].reshape(25, 128)[used] for k in ("itlco", "ituco")])),
        "compton_used": sha(np.stack([T.a[k].reshape(40, 25)[:, used] for k in ("fco", "uico", "fj0")])),
    }
    out["digest_names"] = np.array(list(digests.keys()))
    out["digest_values"] = np.array(list(digests.values()))
    out["used_materials"] = used
    out["sample_rows"] = rows
    out["mfp_a_rows"] = A[rows[rows < nv]].copy()
    out["mfp_b_rows"] = B[rows[rows < nv]].copy()
    out["woodcock_rows"] = T.a["mfp_woodcock"].reshape(nv, 2)[rows[rows < nv - 1]].copy()
    # tallies
    nb = NBATCH[name]
    names = []
    with tempfile.TemporaryDirectory() as rep:
        for p in range(nproj):
            seed = 42 + 1000 * p
            img = ref.track(p, seed, 0, nb, HPT)
            i, v = sparse(img)
            out[f"ref_idx_p{p}"], out[f"ref_val_p{p}"] = i, v
            img_pm, cnt = T.track(p, seed, 0, nb, HPT, ol.MATH_PORTABLE)
            i, v = sparse(img_pm)
            out[f"portable_idx_p{p}"], out[f"portable_val_p{p}"] = i, v
            img_lm, _ = T.track(p, seed, 0, nb, HPT, ol.MATH_LIBM)
            assert np.array_equal(img_lm, img), f"oracle(libm) != reference on {name} p{p}"
            with quiet():
                ref.lib.ref_report((rep + "/projection").encode(), p, nb * HPT, 1.0)
            if p == nproj - 1:
                out["counters_names"] = np.array(list(cnt.as_dict().keys()))
                out["counters_last_projection"] = np.array(list(cnt.as_dict().values()), dtype=np.int64)
        names = sorted(os.listdir(rep), key=lambda f: os.path.getmtime(os.path.join(rep, f)))
        out["file_names"] = np.array(names)
        # ASCII formatting KAT: data lines (non-comment) of the LAST projection, first 3 detector rows + last row
        lines = [l for l in open(os.path.join(rep, names[-1])).read().split("\n")]
        data = [l for l in lines if not l.startswith("#")]
        nx = int(T.detector[0]["num_pixels"][0])
        out["ascii_first_rows"] = np.array(data[: 3 * (nx + 1)])
        out["ascii_num_lines"] = len(lines)
        out["ascii_comment_tail"] = np.array([l for l in lines if l.startswith("#")][-5:])
    out["nbatch_hpt"] = np.array([nb, HPT])
    # dose tallies (accumulated by the reference over the projections tracked above)
    roi, ref_vox, ref_mat = ref.dose()
    if roi[1] > -1 or int(ref.scalars["flag_material_dose"]) == 1:
        out["dose_roi"] = np.array(roi)
        out["dose_materials_ref"] = ref_mat
        for mode, tag in ((ol.MATH_LIBM, "libm"), (ol.MATH_PORTABLE, "portable")):
            vox, mat = T.enable_dose(roi if roi[1] > -1 else None, True)
            for p in range(nproj):
                T.track(p, 42 + 1000 * p, 0, nb, HPT, mode)
            if tag == "libm":
                assert np.array_equal(mat, ref_mat), f"oracle(libm) material dose != reference on {name}"
                if roi[1] > -1:
                    assert np.array_equal(vox.reshape(-1, 2), ref_vox), f"oracle(libm) voxel dose != reference on {name}"
            else:
                out["dose_materials_portable"] = mat.copy()
            if roi[1] > -1:
                i, v = sparse(vox.reshape(-1))
                out[f"dose_voxels_{tag}_idx"], out[f"dose_voxels_{tag}_val"] = i, v
                out["dose_voxels_shape"] = np.array(vox.shape)
        T.ct.voxels_edep, T.ct.materials_dose = None, None
        with tempfile.TemporaryDirectory() as rep:
            with capture(rep + "/stdout.txt"):
                ref.report_dose(rep + "/dose.dat", nb * HPT, 1.0)
            out["dose_stdout_rows"] = np.array([l for l in open(rep + "/stdout.txt").read().split("\n") if l.startswith("\t")])
            if roi[1] > -1:
                lines = open(rep + "/dose.dat").read().split("\n")
                sep = max(i for i, l in enumerate(lines) if l.startswith("# ====="))
                out["dose_file_body"] = np.array(lines[sep + 1:])
                out["dose_raw_sha256"] = np.array([hashlib.sha256(open(rep + "/dose.dat" + sfx, "rb").read()).hexdigest() for sfx in (".raw", "_2sigma.raw")])
        print(f"   dose: ROI {roi}, material counters {int(ref_mat[:, 0].sum())}, voxel words {0 if ref_vox is None else int(np.count_nonzero(ref_vox))}")
    np.savez_compressed(GOLD / f"case_{name}.npz", **out)
    nz = sum(len(out[f"ref_idx_p{p}"]) for p in range(nproj))
    print(f"case {name}: {nproj} projection(s), {nb*HPT} histories each, {nz} non-zero tally words, files {names}")
    return T


def main():
    if not ol.reference_available():
        raise SystemExit("oracle/_ref is missing: run `make -C oracle ref` in the development container")
    GOLD.mkdir(parents=True, exist_ok=True)
    ref = ol.Reference()
    only = sys.argv[1:]
    if not only:
        rng_kat(ref)
    with tempfile.TemporaryDirectory() as wd:
        wd = Path(wd)
        for name in (only or cases.CASES):
            T = case_fixture(ref, name, wd)
            if name == "catphan64" and not only:
                physics_kat(ref, T)


if __name__ == "__main__":
    main()

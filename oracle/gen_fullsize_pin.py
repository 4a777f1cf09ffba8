#!/usr/bin/env python3
"""Pin the host tables of the three BENCH-SIZE workloads against the REAL reference (oracle/_ref).

Test infrastructure; runs only in the development container (the reference does not exist on the GPU box).
Usage:  make -C oracle ref && python oracle/gen_fullsize_pin.py [catphan cirs thorax]

Why: every GPU parity test feeds the oracle the engine's own host tables (tests/parity.py).  Those tables are pinned
against the reference on the eleven small cases (tests/test_host_tables.py); at bench size the volume reaches the engine
through the binary sidecar `geometry.voxbin`, a path whose equality with the text parse was only tested at toy size.  Here
the reference's own `load_voxels` / `load_material` (MC-GPU_v1.3.cu:1996-2443) parse the TEXT files that
`bench.build_workload` writes -- 134 M lines for the 512^3 Catphan -- and the SHA-256 of what they produce is committed as
`tests/golden/fullsize_ref_pin.json`.  tests/test_gpu_fullsize.py then requires the engine's tables, loaded the way the
bench loads them (sidecar), to hash to the same digests.

With --tallies (round 5) the reference then also TRACKS at that size -- BASELINE config 1 at its stated shape for the Catphan
(projection 0, seed 42, 66667 batches x 150 = 10 000 050 histories, MC-GPU_v1.3.cu:823-841), 512 batches for the two tissue
volumes -- and `tests/golden/fullsize_tally_pin.json` + `fullsize_tally_diff_<workload>.npz` keep: class sums, non-zero words
and SHA-256 of the reference's image, the same of the portable restatement's image on the same batches, whether the libm
restatement equals the reference bit for bit, and the few tally words in which reference and portable image differ (index +
the reference's value: last-bit differences of logf).  The GPU test patches those words into the COMPAT kernel's image and
requires the REFERENCE's digest: the chain GPU -> oracle/_ref holds at 512^3.

What is pinned per workload: voxel_mat_dens (float2 {material + 0.0001, density} per voxel, the reference's layout),
density_max, the Woodcock table without its uninitialised last entry (DESIGN.md deviation 1), a/b mean-free-path tables on
the used-material columns, and the pose structs of all 894 projections.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))

import cases  # noqa: E402
import oracle_lib as ol  # noqa: E402
import bench  # noqa: E402

OUT = ROOT / "tests" / "golden" / "fullsize_ref_pin.json"
TALLY_OUT = ROOT / "tests" / "golden" / "fullsize_tally_pin.json"
# projection, seed, batches, histories per thread (the reference's launch shape, MC-GPU_v1.3.cu:823-841)
TALLY_SHAPES = {"catphan": (0, 42, 66667, 150), "thorax": (223, 42, 512, 150), "cirs": (600, 42, 512, 150)}


def image_summary(img, npix) -> dict:
    c = img.reshape(4, npix)
    return {"class_sums": [int(c[k].sum(dtype=np.uint64)) for k in range(4)], "nonzero_words": int(np.count_nonzero(img)), "sha256": sha(img)}


def tally_pin(ref, T, wl) -> dict:
    """Track with the reference itself and with the restatement (libm and portable math) on the reference's own tables."""
    p, seed, nb, hpt = TALLY_SHAPES[wl]
    t0 = time.time()
    img_ref = ref.track(p, seed, 0, nb, hpt).copy()
    t1 = time.time()
    img_libm, _ = T.track(p, seed, 0, nb, hpt, ol.MATH_LIBM, n_threads=8)
    img_port, _ = T.track(p, seed, 0, nb, hpt, ol.MATH_PORTABLE, n_threads=8)
    npix = img_ref.size // 4
    diff = np.flatnonzero(img_ref != img_port).astype(np.uint32)
    np.savez_compressed(ROOT / "tests" / "golden" / f"fullsize_tally_diff_{wl}.npz", index=diff, reference_value=img_ref[diff])
    out = {"projection": p, "seed": seed, "batches": nb, "histories_per_thread": hpt, "histories": nb * hpt,
           "reference": image_summary(img_ref, npix), "portable": image_summary(img_port, npix),
           "libm_restatement_equals_reference": bool(np.array_equal(img_ref, img_libm)),
           "words_reference_differs_from_portable": int(diff.size), "reference_seconds_one_core": round(t1 - t0, 1)}
    print(f"{wl}: reference tracked {nb * hpt} histories in {t1 - t0:.0f} s; libm restatement equal: {out['libm_restatement_equals_reference']}; "
          f"{diff.size} of {out['reference']['nonzero_words']} non-zero words differ from the portable image", flush=True)
    return out


def sha(a) -> str:
    h = hashlib.sha256()
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    for k in range(0, a.size, 1 << 28):
        h.update(a[k:k + (1 << 28)].tobytes())
    return h.hexdigest()


def digests(tab: dict, nv: int, used, nproj: int) -> dict:
    """Digests of a table set given as flat little-endian arrays under the reference's names (shared with the test)."""
    A = np.asarray(tab["mfp_a"]).view("<f4").reshape(nv, 25, 3)[:, used]
    B = np.asarray(tab["mfp_b"]).view("<f4").reshape(nv, 25, 3)[:, used]
    W = np.asarray(tab["mfp_woodcock"]).view("<f4").reshape(nv, 2)[: nv - 1]
    return {"voxel_mat_dens": sha(tab["voxel_mat_dens"]), "density_max": sha(np.asarray(tab["density_max"]).view("<f4")[:22]),
            "woodcock_but_last": sha(W), "mfp_a_used": sha(A), "mfp_b_used": sha(B),
            "source_data": sha(np.asarray(tab["source_data"]).view(np.uint8)[: 80 * nproj]),
            "detector_data": sha(np.asarray(tab["detector_data"]).view(np.uint8)[: 100 * nproj])}


def main():
    if not ol.reference_available():
        raise SystemExit("oracle/_ref is missing: run `make -C oracle ref` in the development container")
    eng = cases.pkg.engine
    eng.load_library()
    pins = json.loads(OUT.read_text()) if OUT.exists() else {}
    tallies = "--tallies" in sys.argv
    tally_pins = json.loads(TALLY_OUT.read_text()) if TALLY_OUT.exists() else {}
    for wl in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["cirs", "thorax", "catphan"]):
        with tempfile.TemporaryDirectory(dir=os.environ.get("PIN_TMP", "/tmp")) as wd:
            wd = Path(wd)
            t0 = time.time()
            inp = bench.build_workload(wd, wl, int(1e8), 894, eng)
            t1 = time.time()
            ref = ol.Reference()
            saved = os.dup(1)
            null = os.open(os.devnull, os.O_WRONLY)
            sys.stdout.flush()
            os.dup2(null, 1)
            try:
                ref.load(inp)  # the reference parses the text voxel file (it knows nothing of the sidecar)
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(null)
                os.close(saved)
            t2 = time.time()
            T = ref.tables()
            used = np.flatnonzero(T.a["noscco"])
            tab = {k: T.a[k] for k in ("voxel_mat_dens", "mfp_a", "mfp_b", "mfp_woodcock", "source_data", "detector_data")}
            tab["density_max"] = ref.get("density_max", "<f4")
            d = digests(tab, T.num_values, used, int(ref.scalars["num_projections"]))
            pins[wl] = {"num_voxels": [int(v) for v in T.num_voxels], "num_projections": int(ref.scalars["num_projections"]),
                        "used_materials": [int(u) for u in used], "sha256": d,
                        "text_voxel_file_bytes": (wd / "geometry.vox").stat().st_size}
            print(f"{wl}: workload written in {t1 - t0:.0f} s, reference load {t2 - t1:.0f} s, voxel_mat_dens {d['voxel_mat_dens'][:16]}...", flush=True)
            if tallies:
                tally_pins[wl] = tally_pin(ref, T, wl)
                TALLY_OUT.write_text(json.dumps(tally_pins, indent=1, sort_keys=True) + "\n")
            del ref, T, tab
        OUT.write_text(json.dumps(pins, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()

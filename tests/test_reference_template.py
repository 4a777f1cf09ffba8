"""Row b (the boundary) on the REFERENCE'S OWN WIRE TEXT: input files rendered from the reference's Jinja2
template `cbctmc/assets/templates/mcgpu_input.jinja2` with the parameter dictionary of `MCSimulation.create_mcgpu_input`
(`cbctmc/mc/simulation.py:314-346`, restated below: mm -> cm rounded to six digits, gpu id -1 for several GPUs, YES/NO for explicit
angles), loaded by the engine's host model (`mcgpu_create(device = -1)`) and by the reference build itself (`oracle/_ref`):
source / detector structs byte for byte, output file names, trajectory scalars.  The template is read where it lies under
/root/reference (development container; skipped elsewhere); what travels to the GPU box are digests
(tests/golden/reference_template_pin.json, written by this file run as a script) that the engine must reproduce from the
package's own formatter.  And the consumer's side of the boundary: `np.loadtxt -> reshape(nz, nx, 4) -> flip -> [:, :crop]`
(`cbctmc/mc/projection.py:42-51`) of an engine-written ASCII projection equals `mcgpu_finalize_projection_host`."""
import hashlib
import json
import os
from pathlib import Path

import numpy as np
import pytest

import cases

TEMPLATE = Path("/root/reference/cbctmc/assets/templates/mcgpu_input.jinja2")
PIN = cases.GOLDEN / "reference_template_pin.json"

# what MCSimulation.run_simulation hands to create_mcgpu_input (sim.py:142-166), for three shapes of scan
SCANS = {
    "single_projection": dict(case="water", n_projections=1, angle_between_projections=360.0 / 894, projection_angles=()),
    "trajectory_894": dict(case="water", n_projections=894, angle_between_projections=360.0 / 894, projection_angles=()),
    # 4-D driver: explicit angles, the first one passed twice (sim.py:658-660 "temporary bug fix for 0th projection always at 270deg")
    "explicit_angles_first_doubled": dict(case="slab_angles", n_projections=894, angle_between_projections=360.0 / 894,
                                          projection_angles=(123.489933, 123.489933, 124.295302, 131.543624, 200.0, 359.597315)),
}


def reference_parameters(geometry_file, material_files, spectrum, source_position_mm, output_folder, n_histories, projection_angles, n_projections,
                         angle_between_projections, gpu_ids=(0,)):
    """The template context of sim.py:314-346 for the reference's default scan (cbctmc/defaults.py:51-96)."""
    d = cases.pkg.defaults.DEFAULTS
    cm = lambda v: round(v / 10.0, 6)  # noqa: E731
    return {
        "gpu_id": -1 if len(gpu_ids) > 1 else gpu_ids[0], "angle_between_projections": angle_between_projections,
        "detector_size_x": cm(d.detector_size[0]), "detector_size_y": cm(d.detector_size[1]),
        "detector_lateral_displacement": cm(d.detector_lateral_displacement), "material_filepaths": [str(p) for p in material_files],
        "n_detector_pixels_x": d.n_detector_pixels[0], "n_detector_pixels_y": d.n_detector_pixels[1], "n_histories": n_histories,
        "specify_projection_angles": "YES" if projection_angles else "NO", "projection_angles": projection_angles, "n_projections": n_projections,
        "output_folder": str(output_folder), "random_seed": d.random_seed,
        "source_polar_aperture_1": d.source_polar_aperture[0], "source_polar_aperture_2": d.source_polar_aperture[1],
        "source_azimuthal_aperture": d.source_azimuthal_aperture,
        "source_direction_cosine_u": d.source_direction_cosines[0], "source_direction_cosine_v": d.source_direction_cosines[1],
        "source_direction_cosine_w": d.source_direction_cosines[2],
        "source_position_x": cm(source_position_mm[0]), "source_position_y": cm(source_position_mm[1]), "source_position_z": cm(source_position_mm[2]),
        "source_to_detector_distance": cm(d.source_to_detector_distance), "source_to_isocenter_distance": cm(d.source_to_isocenter_distance),
        "voxel_geometry_filepath": str(geometry_file), "xray_spectrum_filepath": str(spectrum),
    }


def scan_inputs(name, folder, template_text=None):
    """(input written by the package's formatter, input rendered from the reference's template or None), same scan, same geometry file."""
    s = SCANS[name]
    factory, _ = cases.CASES[s["case"]]
    geo = factory()
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    sim = cases.simulation.MCSimulation(geo, cases.material_files(), cases.spectrum_file(), n_histories=int(2.4e9), n_projections=s["n_projections"],
                                        angle_between_projections=s["angle_between_projections"], projection_angles=list(s["projection_angles"]))
    mine = sim.prepare_simulation(folder, compress_geometry=True)
    rendered = None
    if template_text is not None:
        import jinja2
        d = cases.pkg.defaults.DEFAULTS
        src = cases.simulation.source_position_for(geo.image_size, d.source_to_isocenter_distance)
        ctx = reference_parameters(folder / "geometry.vox.gz", cases.material_files(), cases.spectrum_file(), src, folder, int(2.4e9),
                                   list(s["projection_angles"]), s["n_projections"], s["angle_between_projections"])
        rendered = folder / "input_from_reference_template.in"
        rendered.write_text(jinja2.Environment().from_string(template_text).render(ctx))
    return mine, rendered


def facts(ctx):
    n = ctx.num_projections
    return {"num_projections": n, "source_data": hashlib.sha256(ctx.host_table("source_data").tobytes()).hexdigest(),
            "detector_data": hashlib.sha256(ctx.host_table("detector_data").tobytes()).hexdigest(),
            "file_names": hashlib.sha256("\n".join(os.path.basename(ctx.projection_file_name(p)) for p in range(n)).encode()).hexdigest(),
            "first_file_names": [os.path.basename(ctx.projection_file_name(p)) for p in range(min(n, 3))],
            "scalars": {k: ctx.geti(k) for k in ("total_histories", "seed", "gpu_id", "threads_per_block", "histories_per_thread", "enable_specific_angles")}}


@pytest.mark.skipif(not TEMPLATE.exists(), reason="the reference tree is not present (GPU box): the committed digests stand in for it")
@pytest.mark.parametrize("name", list(SCANS))
def test_engine_reads_the_reference_template_like_the_reference_build(name, engine, tmp_path):
    pytest.importorskip("jinja2")
    import oracle_lib as ol
    mine, rendered = scan_inputs(name, tmp_path, TEMPLATE.read_text())
    with engine.create(rendered, device=-1) as a, engine.create(mine, device=-1) as b:
        fa, fb = facts(a), facts(b)
        assert fa == fb, "the package's formatter and the reference's template describe different scans"
        assert fa == json.loads(PIN.read_text())[name], "tests/golden/reference_template_pin.json is stale: python tests/test_reference_template.py"
        src, det = a.host_table("source_data"), a.host_table("detector_data")
        names = [os.path.basename(a.projection_file_name(p)) for p in range(a.num_projections)]
        assert all(cases.simulation.PROJECTION_FILE_PATTERN.match(n) for n in names)
        if SCANS[name]["projection_angles"]:
            assert names[0] == names[1], "the doubled first angle gives the same file twice (the reference's 4-D driver relies on it)"
    if not ol.reference_available():
        pytest.skip("oracle/_ref not built")
    ref = ol.Reference().load(rendered)
    n = int(ref.scalars["num_projections"])
    assert n == fa["num_projections"]
    assert np.array_equal(ref.get("source_data")[: 80 * n], src) and np.array_equal(ref.get("detector_data")[: 100 * n], det)
    for k, v in fa["scalars"].items():
        assert int(ref.scalars[k]) == v, k


@pytest.mark.parametrize("name", list(SCANS))
def test_package_formatter_reproduces_the_pinned_reference_scans(name, engine, tmp_path):
    """Runs everywhere: the digests were made from the reference's template (test above, development container)."""
    mine, _ = scan_inputs(name, tmp_path)
    with engine.create(mine, device=-1) as ctx:
        assert facts(ctx) == json.loads(PIN.read_text())[name]


def test_reference_consumer_recipe_on_an_engine_written_projection(engine, case_dir, tmp_path):
    """proj.py:42-51 (`_read_raw`): loadtxt -> float32 -> reshape(nz, nx, 4) -> flip(axis 0) -> [:, :crop] on the ASCII file the engine
    writes, against mcgpu_finalize_projection_host on the same tally: total / unscattered / scattered planes identical."""
    rng = np.random.default_rng(5)
    with engine.create(case_dir("catphan64"), device=-1) as ctx:
        nz, nx = ctx.detector_shape
        image = np.zeros((4, nz, nx), dtype=np.uint64)
        hit = rng.random((4, nz, nx)) < 0.4
        image[hit] = rng.integers(1, 2 ** 40, size=int(hit.sum()), dtype=np.uint64)
        histories, crop = 300_000, 128
        f = ctx.write_projection(0, image, histories, seconds=1.0, file_name=str(tmp_path / "projection_0000"))
        data = np.loadtxt(f, dtype=np.float64).astype(np.float32)
        data = np.flip(data.reshape(nz, nx, 4), axis=0)[:, :crop]
        planes = ctx.finalize_host(image, histories, crop_nx=crop)
        # projections_to_itk (proj.py:131-137): total = data.sum(-1), unscattered = data[..., 0], scattered = data[..., 1:].sum(-1)
        assert np.array_equal(planes[0], data.sum(axis=-1))
        assert np.array_equal(planes[1], data[..., 0])
        assert np.array_equal(planes[2], data[..., 1:].sum(axis=-1))
        assert planes[0].max() > 0


if __name__ == "__main__":  # regenerate the pin (development container: needs /root/reference and jinja2)
    import tempfile
    eng = cases.pkg.engine
    out = {}
    for name in SCANS:
        with tempfile.TemporaryDirectory() as tmp:
            _, rendered = scan_inputs(name, tmp, TEMPLATE.read_text())
            with eng.create(rendered, device=-1) as ctx:
                out[name] = facts(ctx)
    PIN.write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", PIN)

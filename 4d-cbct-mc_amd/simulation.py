"""`MCSimulation` host-side mirror: writes the MC-GPU `.in` file and drives the engine.

Mirrors `cbctmc/mc/simulation.py`: `create_mcgpu_input` (:288-357; mm -> cm with round(x/10, 6)),
source position `(Sx/2, Sy/2 - SID, Sz/2)` (:130-136), output naming and the `run_simulation`
flow (:370-427) -- with the docker/mpirun launch (:187-198) replaced by an in-process call into
the engine's C ABI (or a subprocess of the `MC-GPU_v1.3.x` executable).  The `.in` text is
produced by a formatter of our own that emits the section markers and value order the engine
parser expects.
"""
from __future__ import annotations

import re
from pathlib import Path
from typing import Sequence, Tuple

from .defaults import DEFAULTS


def create_mcgpu_input(
    voxel_geometry_filepath,
    material_filepaths: Sequence,
    xray_spectrum_filepath,
    source_position: Tuple[float, float, float],
    output_folder,
    n_histories: int = DEFAULTS.n_histories,
    projection_angles: Sequence[float] = (),
    n_projections: int = DEFAULTS.n_projections,
    angle_between_projections: float = DEFAULTS.angle_between_projections,
    source_direction_cosines=DEFAULTS.source_direction_cosines,
    source_polar_aperture=DEFAULTS.source_polar_aperture,
    source_azimuthal_aperture: float = DEFAULTS.source_azimuthal_aperture,
    n_detector_pixels=DEFAULTS.n_detector_pixels,
    detector_size=DEFAULTS.detector_size,
    detector_lateral_displacement: float = DEFAULTS.detector_lateral_displacement,
    source_to_detector_distance: float = DEFAULTS.source_to_detector_distance,
    source_to_isocenter_distance: float = DEFAULTS.source_to_isocenter_distance,
    random_seed: int = DEFAULTS.random_seed,
    gpu_ids: Sequence[int] = (0,),
    threads_per_block: int = DEFAULTS.threads_per_block,
    histories_per_thread: int = DEFAULTS.histories_per_thread,
    tally_material_dose: bool = False,
    tally_voxel_dose: bool = False,
    dose_roi=((1, 1), (1, 1), (1, 1)),
) -> str:
    """Render an MC-GPU input file.  Lengths in mm (converted to cm like the reference)."""
    cm = lambda v: round(v / 10.0, 6)
    gpu_id = -1 if len(gpu_ids) > 1 else gpu_ids[0]
    L = []
    L.append("# >>>> INPUT FILE FOR MC-GPU v1.3 >>>>")
    L.append("")
    L.append("#[SECTION SIMULATION CONFIG v.2009-05-12]")
    L.append(f"{n_histories}  # TOTAL NUMBER OF HISTORIES, OR SIMULATION TIME IN SECONDS IF VALUE < 100000")
    L.append(f"{random_seed}  # RANDOM SEED (ranecu PRNG)")
    L.append(f"{gpu_id}  # GPU NUMBER TO USE, OR TO BE AVOIDED IN MULTI-GPU RUNS")
    L.append(f"{threads_per_block}  # GPU THREADS PER BLOCK (multiple of 32)")
    L.append(f"{histories_per_thread}  # SIMULATED HISTORIES PER GPU THREAD")
    L.append("")
    L.append("#[SECTION SOURCE v.2011-07-12]")
    L.append(f"{xray_spectrum_filepath}  # X-RAY ENERGY SPECTRUM FILE")
    L.append(f"{cm(source_position[0])} {cm(source_position[1])} {cm(source_position[2])}  # SOURCE POSITION: X Y Z [cm]")
    L.append(f"{source_direction_cosines[0]} {source_direction_cosines[1]} {source_direction_cosines[2]}  # SOURCE DIRECTION COSINES: U V W")
    L.append(f"{source_polar_aperture[0]} {source_polar_aperture[1]} {source_azimuthal_aperture}  # FAN BEAM APERTURES PHI1 PHI2 THETA [degrees]")
    L.append("")
    L.append("#[SECTION IMAGE DETECTOR v.2009-12-02]")
    L.append(f"{output_folder}/projection  # OUTPUT IMAGE FILE NAME")
    L.append(f"{n_detector_pixels[0]} {n_detector_pixels[1]}  # NUMBER OF PIXELS IN THE IMAGE: Nx Nz")
    L.append(f"{cm(detector_size[0])} {cm(detector_size[1])}  # IMAGE SIZE (width, height): Dx Dz [cm]")
    L.append(f"{cm(source_to_detector_distance)}  # SOURCE-TO-DETECTOR DISTANCE")
    L.append(f"{cm(detector_lateral_displacement)}  # LATERAL DETECTOR DISPLACEMENT (along x axis [cm])")
    L.append("")
    L.append("#[SECTION ANGLES OF PROJ v.2023-09-06]")
    L.append(("YES" if len(projection_angles) else "NO") + "  # DEFINE ANGLES SPECIFICALLY? [YES/NO]")
    for i, a in enumerate(projection_angles, 1):
        L.append(f"{a}  # PROJECTION ANGLE {i}")
    L.append("")
    L.append("#[SECTION CT SCAN TRAJECTORY v.2011-10-25]")
    L.append(f"{n_projections}  # NUMBER OF PROJECTIONS")
    L.append(f"{angle_between_projections}  # ANGLE BETWEEN PROJECTIONS [degrees]")
    L.append("0.0 5000.0  # ANGLES OF INTEREST")
    L.append(f"{cm(source_to_isocenter_distance)}  # SOURCE-TO-ROTATION AXIS DISTANCE")
    L.append("0.0  # VERTICAL TRANSLATION BETWEEN PROJECTIONS (HELICAL SCAN)")
    L.append("")
    L.append("#[SECTION DOSE DEPOSITION v.2012-12-12]")
    # the reference template hard-codes NO/NO (mcgpu_input.jinja2:37-38); the engine implements both tallies
    L.append(("YES" if tally_material_dose else "NO") + "  # TALLY MATERIAL DOSE? [YES/NO]")
    L.append(("YES" if tally_voxel_dose else "NO") + "  # TALLY 3D VOXEL DOSE? [YES/NO]")
    L.append(f"{output_folder}/dose.dat  # OUTPUT VOXEL DOSE FILE NAME")
    L.append(f"{dose_roi[0][0]} {dose_roi[0][1]}  # VOXEL DOSE ROI: X-index min max (first voxel has index 1)")
    L.append(f"{dose_roi[1][0]} {dose_roi[1][1]}  # VOXEL DOSE ROI: Y-index min max")
    L.append(f"{dose_roi[2][0]} {dose_roi[2][1]}  # VOXEL DOSE ROI: Z-index min max")
    L.append("")
    L.append("#[SECTION VOXELIZED GEOMETRY FILE v.2009-11-30]")
    L.append(f"{voxel_geometry_filepath}  # VOXELIZED GEOMETRY FILE")
    L.append("")
    L.append("#[SECTION MATERIAL FILE LIST v.2009-11-30]")
    for i, m in enumerate(material_filepaths, 1):
        L.append(f"{m}  # MATERIAL FILE {i}")
    L.append("")
    L.append("# >>>> END INPUT FILE >>>>")
    return "\n".join(L) + "\n"


def source_position_for(image_size_mm, source_to_isocenter_distance=DEFAULTS.source_to_isocenter_distance):
    """Focal-spot position the reference derives from the UN-rotated volume size (sim.py:130-136)."""
    return (image_size_mm[0] / 2, image_size_mm[1] / 2 - source_to_isocenter_distance, image_size_mm[2] / 2)


PROJECTION_FILE_PATTERN = re.compile(r"^projection_\d{3}\.\d{6}deg$")  # sim.py:283


class MCSimulation:
    """Same constructor arguments as the reference's `BaseMCSimulation.__init__` (sim.py:40-70)."""

    def __init__(self, geometry, material_filepaths, xray_spectrum_filepath, n_histories=DEFAULTS.n_histories,
                 projection_angles=(), n_projections=DEFAULTS.n_projections,
                 angle_between_projections=DEFAULTS.angle_between_projections,
                 source_direction_cosines=DEFAULTS.source_direction_cosines,
                 n_detector_pixels=DEFAULTS.n_detector_pixels, detector_size=DEFAULTS.detector_size,
                 source_to_detector_distance=DEFAULTS.source_to_detector_distance,
                 source_to_isocenter_distance=DEFAULTS.source_to_isocenter_distance,
                 random_seed=DEFAULTS.random_seed, source_polar_aperture=DEFAULTS.source_polar_aperture,
                 source_azimuthal_aperture=DEFAULTS.source_azimuthal_aperture,
                 threads_per_block=DEFAULTS.threads_per_block, histories_per_thread=DEFAULTS.histories_per_thread,
                 tally_material_dose=False, tally_voxel_dose=False, dose_roi=((1, 1), (1, 1), (1, 1))):
        self.geometry = geometry
        self.material_filepaths = list(material_filepaths)
        self.xray_spectrum_filepath = xray_spectrum_filepath
        self.n_histories = int(n_histories)
        self.projection_angles = list(projection_angles)
        self.n_projections = len(self.projection_angles) or n_projections
        self.angle_between_projections = angle_between_projections
        self.source_direction_cosines = source_direction_cosines
        self.n_detector_pixels = n_detector_pixels
        self.detector_size = detector_size
        self.source_to_detector_distance = source_to_detector_distance
        self.source_to_isocenter_distance = source_to_isocenter_distance
        self.random_seed = random_seed
        self.source_polar_aperture = source_polar_aperture
        self.source_azimuthal_aperture = source_azimuthal_aperture
        self.threads_per_block = threads_per_block
        self.histories_per_thread = histories_per_thread
        self.tally_material_dose = tally_material_dose
        self.tally_voxel_dose = tally_voxel_dose
        self.dose_roi = dose_roi

    def prepare_simulation(self, output_folder, geometry_output_folder=None, output_suffix="", gpu_ids=(0,),
                           force_geometry_recompile=False, compress_geometry=True, engine=None) -> Path:
        """Write geometry<suffix>.vox.gz and input<suffix>.in (sim.py:95-174); returns the input path."""
        output_folder = Path(output_folder)
        geometry_output_folder = Path(geometry_output_folder or output_folder)
        output_folder.mkdir(parents=True, exist_ok=True)
        geometry_output_folder.mkdir(parents=True, exist_ok=True)
        gpu_ids = (gpu_ids,) if isinstance(gpu_ids, int) else tuple(gpu_ids)
        input_filepath = output_folder / f"input{output_suffix}.in"
        geometry_filepath = geometry_output_folder / (f"geometry{output_suffix}.vox" + (".gz" if compress_geometry else ""))
        if not geometry_filepath.exists() or force_geometry_recompile:
            self.geometry.save_mcgpu_geometry(geometry_filepath, compress=compress_geometry, engine=engine)
        text = create_mcgpu_input(
            voxel_geometry_filepath=geometry_filepath, material_filepaths=self.material_filepaths,
            xray_spectrum_filepath=self.xray_spectrum_filepath,
            source_position=source_position_for(self.geometry.image_size, self.source_to_isocenter_distance),
            output_folder=output_folder, n_histories=self.n_histories, projection_angles=self.projection_angles,
            n_projections=self.n_projections, angle_between_projections=self.angle_between_projections,
            source_direction_cosines=self.source_direction_cosines, source_polar_aperture=self.source_polar_aperture,
            source_azimuthal_aperture=self.source_azimuthal_aperture, n_detector_pixels=self.n_detector_pixels,
            detector_size=self.detector_size, source_to_detector_distance=self.source_to_detector_distance,
            source_to_isocenter_distance=self.source_to_isocenter_distance, random_seed=self.random_seed,
            gpu_ids=gpu_ids, threads_per_block=self.threads_per_block, histories_per_thread=self.histories_per_thread,
            tally_material_dose=self.tally_material_dose, tally_voxel_dose=self.tally_voxel_dose, dose_roi=self.dose_roi)
        input_filepath.write_text(text)
        return input_filepath

    def run_simulation(self, output_folder, engine, gpu_ids=(0,), mode="fast", write_projections=True, **prepare_kwargs):
        """Prepare inputs and run every projection on the GPU engine; returns the list of output files."""
        input_filepath = self.prepare_simulation(output_folder, gpu_ids=gpu_ids, engine=engine, **prepare_kwargs)
        gpu_ids = (gpu_ids,) if isinstance(gpu_ids, int) else tuple(gpu_ids)
        ctx = engine.create(str(input_filepath), device=gpu_ids[0])
        try:
            return ctx.run_all(mode=mode, write_projections=write_projections)
        finally:
            ctx.close()

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
                           force_geometry_recompile=False, compress_geometry=True, engine=None, binary_sidecar=False) -> Path:
        """Write geometry<suffix>.vox.gz and input<suffix>.in (sim.py:95-174); returns the input path."""
        output_folder = Path(output_folder)
        geometry_output_folder = Path(geometry_output_folder or output_folder)
        output_folder.mkdir(parents=True, exist_ok=True)
        geometry_output_folder.mkdir(parents=True, exist_ok=True)
        gpu_ids = (gpu_ids,) if isinstance(gpu_ids, int) else tuple(gpu_ids)
        input_filepath = output_folder / f"input{output_suffix}.in"
        geometry_filepath = geometry_output_folder / (f"geometry{output_suffix}.vox" + (".gz" if compress_geometry else ""))
        if not geometry_filepath.exists() or force_geometry_recompile:
            self.geometry.save_mcgpu_geometry(geometry_filepath, compress=compress_geometry, engine=engine,
                                              binary_sidecar=binary_sidecar and engine is not None)
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

    _AIR_SIMULATION_FOLDER = "air"  # sim.py:38
    STACK_PIXEL_SPACING = (0.776, 0.776)  # the spacing the reference stamps on its stacks (projection.py:73)

    @staticmethod
    def run_air_simulation(output_folder, engine, material_filepaths, xray_spectrum_filepath, n_histories=int(5e10), gpu_ids=(0,),
                           mode="fast", **sim_kwargs):
        """One air projection for the Beer-Lambert normalisation (sim.py:72-87): a 1-voxel air geometry, 5e10 histories."""
        from .geometry import MCAirGeometry
        folder = Path(output_folder) / MCSimulation._AIR_SIMULATION_FOLDER
        sim = MCSimulation(MCAirGeometry(), material_filepaths, xray_spectrum_filepath, n_histories=n_histories, n_projections=1, **sim_kwargs)
        return sim.run_simulation(folder, engine, gpu_ids=gpu_ids, mode=mode, run_air_simulation=False)

    @staticmethod
    def _already_simulated(output_folder) -> bool:
        return (Path(output_folder) / "projections_total.mha").is_file()  # sim.py:89-93

    def run_simulation(self, output_folder, engine, gpu_ids=(0,), mode="fast", run_air_simulation=False,
                       air_projection_denoise_kernel_size=(10, 10), clean=True, stack_projections=True, force_rerun=False,
                       air_n_histories=int(5e10), **prepare_kwargs):
        """`BaseMCSimulation.run_simulation` (sim.py:370-427) on the in-process engine: the docker/mpirun launch and the
        ASCII -> numpy -> SimpleITK post-processing are replaced by the engine's scan pipeline, which writes
        projections_{total,unscattered,scattered}.mha (and projections_total_normalized.mha with an air scan) directly.
        `clean=False` additionally keeps the reference's per-projection ASCII files.  Returns the scan report."""
        output_folder = Path(output_folder)
        if self._already_simulated(output_folder) and not force_rerun:
            return None
        if run_air_simulation and not stack_projections:
            raise ValueError("Cannot perform air normalization without stacking projections")  # sim.py:244-247
        gpu_ids = (gpu_ids,) if isinstance(gpu_ids, int) else tuple(gpu_ids)
        air_stack = None
        if run_air_simulation:
            kw = dict(n_detector_pixels=self.n_detector_pixels, detector_size=self.detector_size,
                      source_to_detector_distance=self.source_to_detector_distance,
                      source_to_isocenter_distance=self.source_to_isocenter_distance, random_seed=self.random_seed,
                      source_polar_aperture=self.source_polar_aperture, source_azimuthal_aperture=self.source_azimuthal_aperture)
            self.run_air_simulation(output_folder, engine, self.material_filepaths, self.xray_spectrum_filepath, n_histories=air_n_histories,
                                    gpu_ids=gpu_ids, mode=mode, **kw)
            air_stack = output_folder / self._AIR_SIMULATION_FOLDER / "projections_total.mha"
        input_filepath = self.prepare_simulation(output_folder, gpu_ids=gpu_ids, engine=engine, **prepare_kwargs)
        half_fan = DEFAULTS.n_detector_pixels_half_fan[0] if tuple(self.n_detector_pixels) == tuple(DEFAULTS.n_detector_pixels) else 0
        ctx = engine.create(str(input_filepath), device=gpu_ids[0])
        try:
            return ctx.run_scan(mode=mode, crop_nx=half_fan, write_ascii=not clean, write_stacks=stack_projections,
                                output_folder=output_folder, air_stack=air_stack, air_sigma=air_projection_denoise_kernel_size,
                                pixel_spacing=self.STACK_PIXEL_SPACING)
        finally:
            ctx.close()

    @staticmethod
    def postprocess_simulation(folder, engine, n_detector_pixels=DEFAULTS.n_detector_pixels,
                               n_detector_pixels_half_fan=DEFAULTS.n_detector_pixels_half_fan, clean=True, stack_projections=True,
                               air_normalization=True, air_projection_denoise_kernel_size=(10, 10)):
        """`postprocess_simulation` (sim.py:235-277) for a folder of ASCII projection files written by the drop-in
        executable: stacks them (same float32 values, MetaImage writer of the engine) and cleans up."""
        import numpy as np
        folder = Path(folder)
        if air_normalization and not stack_projections:
            raise ValueError("Cannot perform air normalization without stacking projections")
        files = sorted(f for f in folder.iterdir() if PROJECTION_FILE_PATTERN.match(f.name))
        if files and stack_projections:
            nx, nz = n_detector_pixels
            cx = n_detector_pixels_half_fan[0] if n_detector_pixels_half_fan else nx
            writers = [engine.StackWriter(folder / f"projections_{m}.mha", cx, nz, len(files), MCSimulation.STACK_PIXEL_SPACING)
                       for m in ("total", "unscattered", "scattered")]
            for f in files:
                data = np.loadtxt(f, dtype=np.float64).astype(np.float32).reshape(nz, nx, 4)  # projection.py:42-51
                data = np.flip(data, axis=0)[:, :cx]
                writers[0].append(data.sum(axis=-1))
                writers[1].append(data[..., 0])
                writers[2].append(data[..., 1:].sum(axis=-1))
            for w in writers:
                w.finish(replace_zeros=True)
            if air_normalization:
                engine.normalize_stack(folder / "projections_total.mha", folder / MCSimulation._AIR_SIMULATION_FOLDER / "projections_total.mha",
                                       folder / "projections_total_normalized.mha", sigma=air_projection_denoise_kernel_size,
                                       spacing=MCSimulation.STACK_PIXEL_SPACING)
        if clean:
            for f in files:
                f.unlink()


class MCSimulation4D:
    """`MCSimulation4D` (sim.py:430-710) on ONE resident engine context.

    The reference launches the engine once per unique respiratory state: warp the geometry on the CPU, write a
    `.vox.gz`, start a container, parse 134 M text lines, simulate that state's projections (the first angle twice,
    sim.py:658-660), and finally read every ASCII file back to stack them.  Here the context is created once; per
    state the geometry is warped on the GPU, handed over as arrays, the state's angles are set, and the scan
    pipeline writes its projections straight into their slices of the three shared stacks.

    `correspondence_model` is anything with `predict(np.array([signal, dt_signal])) -> displacement field
    [3, x, y, z]` in voxels (the reference's CorrespondenceModel.predict, sim.py:477)."""

    def __init__(self, correspondence_model, geometry, material_filepaths, xray_spectrum_filepath, n_histories=DEFAULTS.n_histories,
                 n_projections=DEFAULTS.n_projections, frame_rate=15.0, angle_between_projections=DEFAULTS.angle_between_projections,
                 random_seed=DEFAULTS.random_seed, **sim_kwargs):
        self.correspondence_model = correspondence_model
        self.geometry = geometry
        self.material_filepaths = list(material_filepaths)
        self.xray_spectrum_filepath = xray_spectrum_filepath
        self.n_histories = int(n_histories)
        self.n_projections = n_projections
        self.frame_rate = frame_rate
        self.angle_between_projections = angle_between_projections
        self.random_seed = random_seed
        self.sim_kwargs = sim_kwargs

    def warp_geometry(self, ctx, signal: float, dt_signal: float):
        """`_warp_geometry` (sim.py:473-478) + `MCGeometry.warp` (geo.py:386-439), nearest neighbour, air outside, on the GPU."""
        import numpy as np
        from .geometry import MCGeometry
        from .materials import MATERIALS_125KEV, material_number
        field = np.asarray(self.correspondence_model.predict(np.array([signal, dt_signal])), dtype=np.float32)
        if field.ndim == 5:
            field = field[0]
        mats, dens = self.geometry.materials, self.geometry.densities  # [x, y, z]
        # engine arrays are [z][y][x]; the field's components stay (x, y, z)
        u = np.ascontiguousarray(np.transpose(field, (0, 3, 2, 1)))
        m, d = ctx.warp_volume(np.transpose(mats, (2, 1, 0)), np.transpose(dens, (2, 1, 0)), u, material_number("air"), MATERIALS_125KEV["air"])
        return MCGeometry(np.transpose(m, (2, 1, 0)), np.transpose(d, (2, 1, 0)), self.geometry.image_spacing)

    def apply_state(self, ctx, engine, signal: float, dt_signal: float):
        """Geometry of one respiratory state on the resident context: warped on the device from the predicted field
        (mcgpu_warp_geometry: nothing but the field crosses PCIe); volumes that are not palette volumes take the route
        through host arrays (warp_volume + set_geometry)."""
        import numpy as np
        field = np.asarray(self.correspondence_model.predict(np.array([signal, dt_signal])), dtype=np.float32)
        if field.ndim == 5:
            field = field[0]
        try:
            ctx.warp_geometry(field, frame="geometry")
        except engine.EngineError as e:
            if e.code != -5:
                raise
            ctx.set_geometry(self.warp_geometry(ctx, signal, dt_signal))

    def run_simulation(self, respiratory_signal, respiratory_signal_quantization, output_folder, engine, gpu_ids=(0,), mode="fast",
                       run_air_simulation=False, air_projection_denoise_kernel_size=(10, 10), air_n_histories=int(5e10), start_angle=270.0,
                       force_rerun=False):
        import numpy as np
        from .respiratory import RespiratorySignal
        output_folder = Path(output_folder)
        if MCSimulation._already_simulated(output_folder) and not force_rerun:
            return None
        output_folder.mkdir(parents=True, exist_ok=True)
        gpu_ids = (gpu_ids,) if isinstance(gpu_ids, int) else tuple(gpu_ids)
        sig = respiratory_signal.resample(self.frame_rate)  # one signal value per projection (sim.py:557-563)
        signal, dt_signal = sig.signal[: self.n_projections], sig.dt_signal[: self.n_projections]
        np.savetxt(output_folder / "signal.txt", np.stack((signal, dt_signal)).T, fmt="%.6f",
                   header="original respiratory signal and its derivative\nsignal quantization: None\nsignal dt_signal")
        if respiratory_signal_quantization:
            signal = RespiratorySignal.quantize_signal(signal, n_bins=respiratory_signal_quantization)
            dt_signal = RespiratorySignal.quantize_signal(dt_signal, n_bins=respiratory_signal_quantization)
        np.savetxt(output_folder / "signal_quantized.txt", np.stack((signal, dt_signal)).T, fmt="%.6f",
                   header=f"quantized respiratory signal and its derivative\nsignal quantization: {respiratory_signal_quantization} bins\nsignal dt_signal")
        unique = RespiratorySignal.get_unique_signals(signal=signal, dt_signal=dt_signal)
        n_proj = len(signal)
        base = MCSimulation(self.geometry, self.material_filepaths, self.xray_spectrum_filepath, n_histories=self.n_histories,
                            projection_angles=[start_angle, start_angle + self.angle_between_projections],
                            angle_between_projections=self.angle_between_projections, random_seed=self.random_seed, **self.sim_kwargs)
        air_stack = None
        if run_air_simulation:
            MCSimulation.run_air_simulation(output_folder, engine, self.material_filepaths, self.xray_spectrum_filepath, n_histories=air_n_histories,
                                            gpu_ids=gpu_ids, mode=mode, **self.sim_kwargs)
            air_stack = output_folder / MCSimulation._AIR_SIMULATION_FOLDER / "projections_total.mha"
        input_filepath = base.prepare_simulation(output_folder, gpu_ids=gpu_ids, engine=engine, binary_sidecar=True)
        nx_det, nz_det = base.n_detector_pixels
        half_fan = DEFAULTS.n_detector_pixels_half_fan[0] if tuple(base.n_detector_pixels) == tuple(DEFAULTS.n_detector_pixels) else 0
        cx = half_fan or nx_det
        stacks = [engine.StackWriter(output_folder / f"projections_{m}.mha", cx, nz_det, n_proj, MCSimulation.STACK_PIXEL_SPACING)
                  for m in ("total", "unscattered", "scattered")]
        geometries = {}
        ctx = engine.create(str(input_filepath), device=gpu_ids[0])
        try:
            for (s, ds), indices in unique.items():
                self.apply_state(ctx, engine, s, ds)
                angles = [start_angle + i * self.angle_between_projections for i in indices]
                ctx.set_projection_angles(angles[0:1] + angles)  # pose 0 is the input file's: skipped below (sim.py:658-660)
                ctx.run_scan(mode=mode, first_projection=1, num_projections=len(angles), crop_nx=half_fan, write_stacks=False,
                             output_folder=output_folder, shared_stacks=stacks, slice_of_projection=indices)
                for a in angles:
                    geometries[a] = {"signal": float(s), "dt_signal": float(ds), "signal_quantization": respiratory_signal_quantization}
        finally:
            ctx.close()
        for w in stacks:
            w.finish(replace_zeros=True)
        if air_stack is not None:
            engine.normalize_stack(output_folder / "projections_total.mha", air_stack, output_folder / "projections_total_normalized.mha",
                                   sigma=air_projection_denoise_kernel_size, spacing=MCSimulation.STACK_PIXEL_SPACING)
        import yaml
        with open(output_folder / "projection_geometries.yaml", "wt") as f:
            yaml.dump(dict(sorted(geometries.items())), f)
        return {"unique_states": len(unique), "projections": n_proj}

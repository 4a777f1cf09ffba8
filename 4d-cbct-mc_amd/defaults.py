"""Default scan / simulation parameters of the reference (Varian TrueBeam half-fan CBCT).

Values restated from the reference's `cbctmc/defaults.py:23-96` (all lengths in mm, as there).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class VarianScan:
    n_projections: int = 894
    n_detector_pixels: Tuple[int, int] = (1024, 768)
    detector_pixel_size: Tuple[float, float] = (0.388, 0.388)
    detector_lateral_displacement: float = -159.856
    source_to_detector_distance: float = 1500.0
    source_to_isocenter_distance: float = 1000.0
    gantry_rotation_speed: float = 6.0  # deg/s
    frame_rate: float = 15.0  # frames/s


@dataclass(frozen=True)
class MCDefaults:
    # noise-matched to a real Varian scan (cbctmc/defaults.py:51-52)
    n_histories: int = 11_903_320_312
    n_projections: int = 894
    angle_between_projections: float = 360.0 / 894
    # virtual (full-fan sized) detector in pixels / mm
    n_detector_pixels: Tuple[int, int] = (1848, 768)
    n_detector_pixels_half_fan: Tuple[int, int] = (1024, 768)
    detector_size: Tuple[float, float] = (717.024, 297.984)
    detector_pixel_size: Tuple[float, float] = (0.388, 0.388)
    detector_lateral_displacement: float = -159.856
    source_to_detector_distance: float = 1500.0
    source_to_isocenter_distance: float = 1000.0
    random_seed: int = 42
    source_direction_cosines: Tuple[float, float, float] = (0.0, 1.0, 0.0)
    # asymmetric half-fan aperture in degrees (cbctmc/defaults.py:88-92); negative theta = fit detector
    source_polar_aperture: Tuple[float, float] = (1.481720423651376, 13.441979314886868)
    source_azimuthal_aperture: float = -1
    # launch shape hard-coded in the reference input template (mcgpu_input.jinja2:7-8)
    threads_per_block: int = 128
    histories_per_thread: int = 150
    spectrum_name: str = "125kVp_0.89mmTi_varian_norm.spc"


DEFAULTS = MCDefaults()

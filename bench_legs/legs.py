"""Driver-timed legs measured after the timed region at N = 1: COMPAT kernel, pipelined scans, the other workloads, 4-D, FDK."""
from __future__ import annotations

import os
import tempfile
import time
from pathlib import Path

import numpy as np

from .checks import entry_face_deficit
from .common import WORKLOADS, build_workload
from .roofline import roofline_block, timed_launches

def compat_leg(ctx, torch, H, launches=3):
    """COMPAT personality (RANECU leap-frog streams, portable restatement of the reference's arithmetic, bit-identical to the
    oracle's portable mode) timed like the
    FAST steps: same projection schedule, the reference's launch shape for H histories (MC-GPU_v1.3.cu:824-841)."""
    nz, nx = ctx.detector_shape
    image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    batches, hpt, total = ctx.reference_shape(H)
    seed = ctx.geti("seed")
    ctx.launch(0, image.data_ptr(), batches, mode="compat", seed=seed, hpt=hpt, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(launches):
        ctx.clear(image.data_ptr(), stream)
        ctx.launch(((i + 1) * 149) % ctx.num_projections, image.data_ptr(), batches, mode="compat", seed=seed, hpt=hpt, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": total * launches / dt, "unit": "histories/s", "launches": launches, "histories_per_launch": total, "ms_per_launch": dt / launches * 1e3,
            "what": "COMPAT kernel: RANECU streams + a portable restatement of the reference arithmetic (own log/pow/sincos, glibc's expf): tallies bit-identical to the CPU oracle's portable mode (tests/test_gpu_fullsize.py), which differs from the reference build on <= 0.2 % of the tally words (last-bit differences of logf; tests/test_gpu_parity.py::test_compat_kernel_against_the_reference_build_itself)"}

def end_to_end_scan(ctx, H, workdir, n=894, ascii_files=False, chunk=64):
    """The pipelined scan driver (track -> finalize -> pinned copy -> writer thread) over `n` projections with its output on
    disk: the three MetaImage stacks, or (`ascii_files`) the reference's ASCII file per projection -- 63 MB of text each,
    formatted on the device (the unchanged-cbctmc drop-in default).  Per-projection wall time including output.
    The ASCII scan runs in pieces of `chunk` projections whose files are deleted before the next piece starts: 894 files are 56 GB
    of text, more than a scratch disk (or the page cache behind it) should be asked to hold for a measurement."""
    out = workdir / ("scan_ascii" if ascii_files else "scan_out")
    out.mkdir(exist_ok=True)
    crop = 1024 if ctx.detector_shape[1] == 1848 else 0
    first = min(100, max(ctx.num_projections - n, 0))
    pieces = [(first + k, min(chunk, n - k)) for k in range(0, n, chunk)] if ascii_files else [(first, n)]
    tot = {"seconds_total": 0.0, "seconds_kernels": 0.0, "seconds_writer": 0.0, "seconds_after_last_kernel": 0.0}
    k_min, k_max = float("inf"), 0.0
    sizes, written = [], 0
    for p0, m in pieces:
        rep = ctx.run_scan(mode="fast", first_projection=p0, num_projections=m, histories=H, crop_nx=crop, write_stacks=not ascii_files,
                           write_ascii=ascii_files, output_folder=out, pixel_spacing=(0.776, 0.776))
        for k in tot:
            tot[k] += rep[k]
        k_min, k_max = min(k_min, rep["kernel_ms_min"]), max(k_max, rep["kernel_ms_max"])
        for f in out.glob("projections_*.mha"):
            f.unlink()
        if ascii_files:
            files = [Path(ctx.projection_file_name(p)) for p in range(p0, p0 + m)]
            sizes += [f.stat().st_size for f in files if f.exists()]
            written += int(sum(f.exists() for f in files))
            for f in files:
                f.unlink(missing_ok=True)
    res = {"projections": n, "seconds_total": tot["seconds_total"], "ms_per_projection_kernels": tot["seconds_kernels"] / n * 1e3,
           "kernel_ms_min_mean_max_over_the_arc": [k_min, tot["seconds_kernels"] / n * 1e3, k_max],
           "writer_ms_per_projection": tot["seconds_writer"] / n * 1e3, "drain_after_last_kernel_ms": tot["seconds_after_last_kernel"] * 1e3}
    if ascii_files:
        res["scans"] = len(pieces)
        res["file_bytes_mean"] = float(np.mean(sizes)) if sizes else 0.0
        res["files_written"] = written
        res["histories_per_s_with_ascii_files"] = n * H / tot["seconds_total"]
        res["ms_per_projection_with_ascii_files"] = tot["seconds_total"] / n * 1e3
    else:
        res["histories_per_s_with_stacks"] = n * H / tot["seconds_total"]
        res["ms_per_projection_with_stacks"] = tot["seconds_total"] / n * 1e3
    return res


def text_geometry_load(eng, inp, device):
    """What a drop-in user pays who brings only the reference's own voxel file: context creation with the .voxbin sidecar ignored
    (MCGPU_IGNORE_VOXBIN: the text parse of load_voxels, MC-GPU_v1.3.cu:2098-2142 -- 134 M lines for the 512^3 volume -- then the same
    table builders and uploads)."""
    os.environ["MCGPU_IGNORE_VOXBIN"] = "1"
    try:
        t0 = time.perf_counter()
        with eng.create(inp, device=device):
            dt = time.perf_counter() - t0
    finally:
        del os.environ["MCGPU_IGNORE_VOXBIN"]
    return dt


def cirs_4d_leg(c2, torch, H, states=10, projections_per_state=894):
    """Config 5 (cbctmc/mc/simulation.py:527-710): `states` respiratory states of the CIRS phantom, each a 167 MB displacement
    field uploaded and applied ON THE DEVICE (mcgpu_warp_geometry: warp of the index volume, brick grids, object box, majorant),
    followed by `projections_per_state` projections of H histories in the warped geometry."""
    nz, nx = c2.detector_shape
    shape = (c2.geti("num_voxels_y"), c2.geti("num_voxels_x"), c2.geti("num_voxels_z"))  # frame of the MCGeometry arrays (engine.warp_geometry)
    image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed = c2.geti("seed")
    zz = np.linspace(-1, 1, shape[2], dtype=np.float32)[None, None, :]
    field = np.zeros((3,) + shape, np.float32)
    warp_s = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in range(states):
        field[2] = (15.0 * np.sin(2 * np.pi * st / states)) * (1 - zz * zz)  # SI motion up to 15 mm (SURVEY 8d input 4)
        torch.cuda.synchronize()  # the previous state's projections are done before the geometry changes under them
        tw = time.perf_counter()
        c2.warp_geometry(field, frame="geometry")
        warp_s.append(time.perf_counter() - tw)
        for k in range(projections_per_state):
            c2.clear(image.data_ptr(), stream)
            c2.launch((st * projections_per_state + k) % c2.num_projections, image.data_ptr(), H, mode="fast", seed=seed, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c2.warp_geometry(np.zeros((3,) + shape, np.float32), frame="geometry")  # back to the base geometry
    return {"states": states, "projections_per_state": projections_per_state, "field_bytes": int(field.nbytes), "seconds_total": dt,
            "ms_per_state_change": float(np.mean(warp_s[1:]) * 1e3), "ms_first_state_change": warp_s[0] * 1e3,
            "value": states * projections_per_state * H / dt, "unit": "histories/s",
            "what": "device-side respiratory states (field upload + warp + brick grids + majorant) followed by their projections, one stream"}

def fdk_leg(pkg, device, n=894, nu=1024, nv=768, du=0.388, pad=1.0):
    """Config 4's reconstruction (cbctmc/reconstruction/reconstruction.py:22-69: rtkfdk --pad 1 --hann 1 --hannY 1, 464 x 250 x
    464 voxels of 1 mm) through the in-process FDK (csrc/fdk.hip, parity unpinned against RTK): synthetic projections of the
    reference's size, kernel times from HIP events inside the library, wall time including the 2.8 GB upload."""
    recon = pkg.reconstruction
    geo = recon.create_geometry(n, start_angle=90.0)
    u0, v0 = -(nu - 1) / 2 * du, -(nv - 1) / 2 * du
    u = (np.arange(nu, dtype=np.float32) - nu / 2) / nu
    proj = np.empty((n, nv, nu), dtype=np.float32)
    proj[:] = (2.0 * np.sqrt(np.maximum(0.0, 0.16 - u * u)))[None, None, :]  # a cylinder's line integrals (the timing does not depend on the values)
    dim = (464, 250, 464)
    wall, r = None, None
    for rep in range(2):  # the first call pays plan creation and allocations
        t0 = time.perf_counter()
        vol, r = recon.fdk(proj, geo, (du, du), (u0, v0), dim, (1.0, 1.0, 1.0), hann=1.0, hann_y=1.0, pad=pad, gpu_id=device)
        wall = time.perf_counter() - t0
    upd = n * dim[0] * dim[1] * dim[2]
    return {"projections": n, "detector": f"{nu}x{nv}", "volume": "464x250x464", "pad": pad, "ms_filter": r["ms_filter"], "ms_backproject": r["ms_backproject"],
            "ms_kernels": r["ms_filter"] + r["ms_backproject"], "voxel_updates_per_s": upd / (r["ms_backproject"] * 1e-3),
            "wall_s_including_host_transfers": wall, "finite": bool(np.isfinite(vol).all()), "parity": "unpinned against RTK (DESIGN.md 2)"}

def other_workloads(eng, torch, H, projections, device, ceilings=None, scans=True):
    """Configs 3-5 under the driver's clock: the same kernel measurement (8 launches of H histories at angles (i 149) mod 894) on the
    bundled CIRS phantom, on the patient-like thorax and on the same thorax with the voxel-level bone texture of the reference's bone
    mapper -- and, because those kernels vary with the angle, the WHOLE 894-projection scan of each with its stacks on disk
    (`end_to_end`, `sustained_value`: what the metric means by a scan, MC-GPU_v1.3.cu:667) with the fastest / mean / slowest kernel
    of the arc; config 5 as 10 respiratory states x 894 projections."""
    out = {}
    for wl in ("cirs", "thorax", "thorax_textured"):
        t0 = time.perf_counter()
        wd = Path(os.path.join(tempfile.gettempdir(), f"mcgpu_bench_{wl}_512_{projections}"))
        inp = wd / "input.in"
        if not (inp.exists() and (wd / "geometry.voxbin").exists()):
            wd.mkdir(parents=True, exist_ok=True)
            build_workload(wd, wl, H, projections, eng)
        t1 = time.perf_counter()
        with eng.create(inp, device=device) as c2:
            k_ms, k_min, detected = timed_launches(c2, torch, H)
            roof, valu = roofline_block(wl, H, k_ms, ceilings, c2)
            out[wl] = {"value": H / (k_ms * 1e-3), "unit": "histories/s", "kernel_ms_avg": k_ms, "kernel_ms_min": k_min, "launches": 8,
                       "config": WORKLOADS[wl][0], "roofline": {k: roof[k] for k in ("kernel", "frac", "achieved", "traffic", "hbm_counter_frac", "fabric_bytes_per_history",
                                                                                     "l2_hit_rate", "traffic_source", "algorithmic_bytes_per_history")},
                       "valu_issue": valu, "volume_bytes": c2.geti("volume_bytes_device"), "materials_used": c2.geti("num_materials_used"),
                       "detected_energy_units_last_projection": detected,
                       # the same kernel with the reference's three double-precision sub-steps (mode fast64), timed like `value`
                       "value_reference_arithmetic": H / (timed_launches(c2, torch, H, mode="fast64")[0] * 1e-3),
                       # the bit-exact personality on this workload (reference arithmetic, RANECU streams), driver-timed like the rest
                       "compat": {k: v for k, v in compat_leg(c2, torch, H, launches=2).items() if k != "what"},
                       "prepare_inputs_s": t1 - t0, "load_measure_s": time.perf_counter() - t1}
            out[wl]["roofline"]["binding"] = roof.get("binding")
            if scans:
                out[wl]["end_to_end"] = end_to_end_scan(c2, H, wd, n=min(projections, c2.num_projections))
                out[wl]["sustained_value"] = out[wl]["end_to_end"]["histories_per_s_with_stacks"]
            if wl == "thorax":
                out[wl]["entry_face_shell"] = entry_face_deficit(c2)
            if wl == "cirs":
                t4 = time.perf_counter()
                out["cirs_4d"] = cirs_4d_leg(c2, torch, H, projections_per_state=projections if scans else 89)
                out["cirs_4d"]["leg_s"] = time.perf_counter() - t4
    return out

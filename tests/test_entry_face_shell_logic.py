"""The FAST kernel's entry-face shell (track_pool.inc: entry_face_shell) restated in float32 numpy against the reference's arithmetic
(move_to_bbox on all axes, first Woodcock step, locate_voxel) on the same sampled photons -- CPU only; the device code itself is held
against the COMPAT personality in tests/test_gpu_fullsize.py::test_fast_reproduces_the_entry_face_shell."""
import sys
from pathlib import Path

import cases

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))


def test_device_logic_selects_the_photons_the_reference_arithmetic_selects(engine, case_dir):
    """A slab case seen from three angles (rotated poses: oblique entry, the only place the shell exists): the two restatements pick
    exactly the same photons, the device's two-level pre-test skips none of them, and at an oblique pose there are some."""
    import entry_face_quirk as q
    total = 0
    with engine.create(case_dir("slab_angles"), device=-1) as ctx:
        for p in range(ctx.num_projections):
            r = q.shell_rates(ctx, p, n_photons=1_500_000, seed=11 + p)
            assert r["device_only"] == 0 and r["reference_only"] == 0 and r["pretest_misses"] == 0, (p, r)
            assert r["device"] == r["reference"]
            total += r["reference"]
    assert total > 0, "no projection of the case has an entry-face shell: the test would hold vacuously"

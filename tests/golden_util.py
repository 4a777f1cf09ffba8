"""Loading of the committed golden fixtures (tests/golden/*.npz, written by oracle/gen_golden.py)."""
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"


def load(name):
    return np.load(GOLD / name, allow_pickle=False)


def dense(g, prefix, p, size):
    img = np.zeros(size, dtype=np.uint64)
    img[g[f"{prefix}_idx_p{p}"]] = g[f"{prefix}_val_p{p}"]
    return img


def scalars(g):
    return dict(zip([str(s) for s in g["scalars_names"]], [float(v) for v in g["scalars"]]))


def sha(a):
    """SHA-256 of an array's bytes (as oracle/gen_fullsize_pin.py hashes tables and images)."""
    import hashlib
    h = hashlib.sha256()
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    for k in range(0, a.size, 1 << 28):
        h.update(a[k:k + (1 << 28)].tobytes())
    return h.hexdigest()


def fullsize_digests(ctx):
    """SHA-256 digests of a context's host tables in the form oracle/gen_fullsize_pin.py took them from the reference."""

    nv, nproj = ctx.geti("num_energy_values"), ctx.num_projections
    used = np.flatnonzero(ctx.host_table("noscco", "<i4")[:25])
    A = ctx.host_table("mfp_a", "<f4").reshape(nv, 25, 3)[:, used]
    B = ctx.host_table("mfp_b", "<f4").reshape(nv, 25, 3)[:, used]
    W = ctx.host_table("mfp_woodcock", "<f4").reshape(nv, 2)[: nv - 1]
    return {"voxel_mat_dens": sha(ctx.host_table("voxel_mat_dens")), "density_max": sha(ctx.host_table("density_max", "<f4")[:22]),
            "woodcock_but_last": sha(W), "mfp_a_used": sha(A), "mfp_b_used": sha(B),
            "source_data": sha(ctx.host_table("source_data")[: 80 * nproj]), "detector_data": sha(ctx.host_table("detector_data")[: 100 * nproj])}, [int(u) for u in used]


def fullsize_pin(workload):
    import json
    return json.loads((GOLD / "fullsize_ref_pin.json").read_text())[workload]


def fullsize_tally_pin(workload):
    """(pin, index, reference_value): the reference's own tallies at bench size (oracle/gen_fullsize_pin.py --tallies) and the tally
    words in which its image differs from the portable restatement's."""
    import json
    pin = json.loads((GOLD / "fullsize_tally_pin.json").read_text())[workload]
    d = load(f"fullsize_tally_diff_{workload}.npz")
    return pin, d["index"], d["reference_value"]

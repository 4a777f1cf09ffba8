"""Dev check (CPU): how many photons the reference's arithmetic scores as un-attenuated primaries right at the face through which
they enter the voxel volume.  move_to_bbox (K.cu:714-805) puts a photon EPS_SOURCE = 1.5e-5 cm past the face ALONG ITS RAY --
for an oblique ray less than EPS_SOURCE perpendicular to the face -- and locate_voxel (K.cu:1036-1042) calls everything within
EPS_SOURCE of a face "outside": a first Woodcock step shorter than about 1.6e-5 cm leaves the photon in that shell, and it is
tallied at once with its full energy.  The COMPAT kernel, the oracle and FAST's move_to_bbox route reproduce this; FAST's
source_entry (the exterior hop at the source) lands the photon deep inside and does not.
Round 5: the FAST kernel reproduces the shell (track_pool.inc: entry_face_shell).  `shell_rates` below evaluates, on the same sampled
photons, the reference's arithmetic AND a float32 restatement of the device function (entry axis from the reciprocal slab test,
correctly rounded division on that axis only, series for -ln(xi), the two-level pre-test): they must select the same photons
(tests/test_entry_face_shell_logic.py).
Usage: python tools/entry_face_quirk.py <workload dir with input.in> [projection ...]   (float32 emulation of the entry and the first step)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from pathlib import Path
f32 = np.float32


def shell_rates(ctx, p, n_photons=4_000_000, seed=5):
    """(photons entering, selected by the reference's arithmetic, selected by the device logic, device-only, reference-only, lanes the
    device's pre-test would have skipped wrongly) for projection p of a host context."""
    src = np.frombuffer(ctx.host_table('source_data'), dtype=np.float32).reshape(-1, 20)
    wk = ctx.host_table('mfp_woodcock', '<f4').reshape(-1, 2)
    e0, ide = f32(ctx.getf('e0')), f32(ctx.getf('ide'))
    espc = ctx.host_table('espc', '<f4'); nb = ctx.geti('num_spectrum_bins')
    cut = ctx.host_table('espc_cutoff', '<f4'); al = ctx.host_table('espc_alias', '<i2')
    bbox = ctx.host_table('size_bbox', '<f4')
    rng = np.random.default_rng(seed)
    S = src[p]; pos = S[:3]; rot = S[6:15].reshape(3, 3); ctl, phl, dct, dph, mh = S[15:20]
    N = n_photons
    RN = rng.random(N).astype(f32) * f32(nb); ip = RN.astype(int); fr = RN - ip
    b = np.where(fr < cut[ip], ip, al[ip]); E = (espc[b] + rng.random(N).astype(f32) * (espc[b + 1] - espc[b])).astype(f32)
    dz = (ctl + rng.random(N).astype(f32) * dct).astype(f32); phi = (phl + rng.random(N).astype(f32) * dph).astype(f32)
    sth = np.sqrt(f32(1) - dz * dz); dy = sth * np.sin(phi).astype(f32); dx = sth * np.cos(phi).astype(f32)
    ok = np.abs(dz / (dy + f32(1e-7))) <= mh
    d = np.stack([rot[0, 0] * dx + rot[0, 1] * dy + rot[0, 2] * dz, rot[1, 0] * dx + rot[1, 1] * dy + rot[1, 2] * dz,
                  rot[2, 0] * dx + rot[2, 1] * dy + rot[2, 2] * dz], 1).astype(f32)[ok]
    E = E[ok]
    n = len(E); kE = f32(0.000015)
    xi = (np.floor(rng.random(n) * 2 ** 24) * 2.0 ** -24 + 2.0 ** -26).astype(f32)  # the device's deviates: k 2^-24 + 2^-26
    idx = np.floor((E - e0) * ide).astype(int); mfp = (wk[idx, 0] + E * wk[idx, 1]).astype(f32)
    # ---- the reference: move_to_bbox on all three axes (K.cu:714-805), the first Woodcock step, locate_voxel's test (K.cu:1036-1042)
    def entry(pc, dc, size):
        r = np.full(pc.shape, f32(-500000.0), dtype=f32)
        m1 = dc > kE; r[m1] = np.where(pc[m1] > 0, f32(0), kE + (-pc[m1]) / dc[m1])
        m2 = dc < -kE; r[m2] = np.where(pc[m2] < size, f32(0), kE + (size - pc[m2]) / dc[m2])
        return r.astype(f32)
    P = np.broadcast_to(pos, d.shape).astype(f32).copy()
    t = np.maximum(np.maximum(entry(P[:, 0], d[:, 0], bbox[0]), entry(P[:, 1], d[:, 1], bbox[1])), entry(P[:, 2], d[:, 2], bbox[2])).astype(f32)
    P0 = (P + (t[:, None] * d).astype(f32)).astype(f32)
    inside0 = (P0[:, 0] >= 0) & (P0[:, 0] <= bbox[0]) & (P0[:, 1] >= 0) & (P0[:, 1] <= bbox[1]) & (P0[:, 2] >= 0) & (P0[:, 2] <= bbox[2])
    step = (-mfp * np.log(xi.astype(np.float64))).astype(f32)
    Q = (P0 + (step[:, None] * d).astype(f32)).astype(f32)
    out = (Q[:, 1] < kE) | (Q[:, 1] > bbox[1] - kE) | (Q[:, 0] < kE) | (Q[:, 0] > bbox[0] - kE) | (Q[:, 2] < kE) | (Q[:, 2] > bbox[2] - kE)
    ref = out & (step < f32(1e-3)) & inside0
    # ---- the device (track_pool.inc: source_entry + entry_face_shell): entry axis from the reciprocal slab test, that axis only
    inv = (f32(1) / d).astype(f32)
    a = (-P * inv).astype(f32); bb = ((bbox[None, :] - P) * inv).astype(f32)
    nmin = np.minimum(a, bb); v_in = np.maximum(nmin.max(axis=1), 0).astype(f32); v_out = np.maximum(a, bb).min(axis=1)
    ax = np.where(v_in == nmin[:, 0], 0, np.where(v_in == nmin[:, 1], 1, 2))
    r = np.arange(n); pa = P[r, ax]; da = d[r, ax]; sz = bbox[ax]
    num = np.where(da > 0, -pa, sz - pa).astype(f32)
    q = (num.astype(np.float64) / da.astype(np.float64)).astype(f32)  # correctly rounded quotient (Markstein's correction on the device)
    c = (pa + ((kE + q).astype(f32) * da).astype(f32)).astype(f32)
    u = (f32(1) - xi).astype(f32); st = (mfp * (u + (f32(0.5) * u) * u).astype(f32)).astype(f32)
    c = (c + (st * da).astype(f32)).astype(f32)
    dev = np.where(da > 0, c < kE, c > sz - kE) & (v_out >= v_in) & (v_in > 0)
    reach = (u * mfp).astype(f32)  # (the device multiplies by the coarse majorant, which is smaller still)
    pre = (reach <= f32(16.0 * 6.0e-5)) & (reach <= f32(6.0e-5) * np.abs(inv[r, ax]))
    return {"entering": int(inside0.sum()), "reference": int(ref.sum()), "device": int(dev.sum()), "device_only": int((dev & ~ref).sum()),
            "reference_only": int((ref & ~dev).sum()), "pretest_misses": int((dev & ~pre).sum()),
            "energy_weighted_rate": float((E * ref)[inside0].sum() / E[inside0].sum()), "mean_inverse_mfp": float(np.mean(1 / mfp[inside0]))}


if __name__ == "__main__":
    import cases
    eng = cases.pkg.engine
    with eng.create(Path(sys.argv[1]) / 'input.in', device=-1) as ctx:
        for p in ([int(v) for v in sys.argv[2:]] or [150, 600, 223]):
            r = shell_rates(ctx, p)
            print(f"projection {p}: of {r['entering']} entering photons the reference's arithmetic lets {r['reference']} escape at once through the entry face "
                  f"(energy-weighted {r['energy_weighted_rate']:.3e}); the device logic selects {r['device']} ({r['device_only']} it alone, {r['reference_only']} the reference alone, "
                  f"{r['pretest_misses']} missed by its pre-test); mean 1/mfpW {r['mean_inverse_mfp']:.2f} /cm")

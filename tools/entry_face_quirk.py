"""Dev check (CPU): how many photons the reference's arithmetic scores as un-attenuated primaries right at the face through which
they enter the voxel volume.  move_to_bbox (K.cu:714-805) puts a photon EPS_SOURCE = 1.5e-5 cm past the face ALONG ITS RAY --
for an oblique ray less than EPS_SOURCE perpendicular to the face -- and locate_voxel (K.cu:1036-1042) calls everything within
EPS_SOURCE of a face "outside": a first Woodcock step shorter than about 1.6e-5 cm leaves the photon in that shell, and it is
tallied at once with its full energy.  The COMPAT kernel, the oracle and FAST's move_to_bbox route reproduce this; FAST's
source_entry (the exterior hop at the source) lands the photon deep inside and does not.
Usage: python tools/entry_face_quirk.py <workload dir with input.in> [projection ...]   (float32 emulation of the entry and the first step)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
from pathlib import Path
eng = cases.pkg.engine
f32 = np.float32
with eng.create(Path(sys.argv[1]) / 'input.in', device=-1) as ctx:
    src = np.frombuffer(ctx.host_table('source_data'), dtype=np.float32).reshape(-1, 20)
    wk = ctx.host_table('mfp_woodcock','<f4').reshape(-1,2)
    e0, ide = f32(ctx.getf('e0')), f32(ctx.getf('ide'))
    espc = ctx.host_table('espc','<f4'); nb = ctx.geti('num_spectrum_bins')
    cut = ctx.host_table('espc_cutoff','<f4'); al = ctx.host_table('espc_alias','<i2')
    bbox = ctx.host_table('size_bbox','<f4')
    rng = np.random.default_rng(5)
    for p in ([int(v) for v in sys.argv[2:]] or [150, 600, 223]):
        S = src[p]; pos = S[:3]; rot = S[6:15].reshape(3,3); ctl, phl, dct, dph, mh = S[15:20]
        N = 4_000_000
        # energy
        RN = rng.random(N).astype(f32)*f32(nb); ip = RN.astype(int); fr = RN - ip
        b = np.where(fr < cut[ip], ip, al[ip]); E = (espc[b] + rng.random(N).astype(f32)*(espc[b+1]-espc[b])).astype(f32)
        dz = (ctl + rng.random(N).astype(f32)*dct).astype(f32); phi = (phl + rng.random(N).astype(f32)*dph).astype(f32)
        sth = np.sqrt(f32(1)-dz*dz); dy = sth*np.sin(phi).astype(f32); dx = sth*np.cos(phi).astype(f32)
        ok = np.abs(dz/(dy+f32(1e-7))) <= mh
        d = np.stack([rot[0,0]*dx+rot[0,1]*dy+rot[0,2]*dz, rot[1,0]*dx+rot[1,1]*dy+rot[1,2]*dz, rot[2,0]*dx+rot[2,1]*dy+rot[2,2]*dz],1).astype(f32)[ok]; E=E[ok]
        kE = f32(0.000015)
        def entry(pc, dc, size):
            r = np.full(pc.shape, f32(-500000.0), dtype=f32)
            m1 = dc > kE; r[m1] = np.where(pc[m1] > 0, f32(0), kE + (-pc[m1])/dc[m1])
            m2 = dc < -kE; r[m2] = np.where(pc[m2] < size, f32(0), kE + (size-pc[m2])/dc[m2])
            return r.astype(f32)
        P = np.broadcast_to(pos, d.shape).astype(f32).copy()
        ey = entry(P[:,1], d[:,1], bbox[1]); ex = entry(P[:,0], d[:,0], bbox[0]); ez = entry(P[:,2], d[:,2], bbox[2])
        t = np.maximum(np.maximum(ex, ey), ez).astype(f32)
        P = (P + t[:,None]*d).astype(f32)
        inside0 = (P[:,0]>=0)&(P[:,0]<=bbox[0])&(P[:,1]>=0)&(P[:,1]<=bbox[1])&(P[:,2]>=0)&(P[:,2]<=bbox[2])
        idx = ((E - e0)*ide + f32(0.00001)).astype(int); mfpW = (wk[idx,0] + E*wk[idx,1]).astype(f32)
        step = (-mfpW*np.log(rng.random(len(E)).astype(f32)+f32(1e-30))).astype(f32)
        Q = (P + step[:,None]*d).astype(f32)
        out1 = (Q[:,1]<kE)|(Q[:,1]>bbox[1]-kE)|(Q[:,0]<kE)|(Q[:,0]>bbox[0]-kE)|(Q[:,2]<kE)|(Q[:,2]>bbox[2]-kE)
        # only those that are still within a hair of the ENTRY face (not legitimately leaving after a long step)
        near = out1 & (step < f32(1e-3)) & inside0
        print(f"projection {p}: entering {inside0.mean():.4f} of the emitted; escaping at once through the entry face {near[inside0].mean():.3e} of them "
              f"(energy-weighted {(E*near)[inside0].sum()/ (E[inside0]).sum():.3e}); mean 1/mfpW {np.mean(1/mfpW[inside0]):.2f} /cm")

"""Dev estimate (CPU, numpy): would LOCAL Woodcock majorants pay on the thorax workload?

Scheme: bricks whose surroundings hold nothing denser than a class threshold within a safe radius r (Euclidean distance
transform over the brick grid, minus the brick diagonal) fly with that class's majorant and steps capped at r (a capped step is a
move without a collision: exact by memorylessness); everything else flies with the global majorant as now.  A crude photon model
(one energy, isotropic scattering, 15 % absorption) counts flight steps per history under both rules.

usage: python tools/local_majorant_estimate.py thorax|cirs <class thresholds as fractions of the global majorant, comma separated> <smallest radius in mm>
Result (round 5, docs/history.md section 10): 14.3 steps per history with the global majorant (the kernel counts 13.9), 13.5-13.9
with local ones: the ribs every 20 mm and the vessels in the lungs leave almost no brick a bone-free radius of one mean free path.
"""
import sys, numpy as np, importlib.util
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location('pkg', str(ROOT / '4d-cbct-mc_amd' / '__init__.py'), submodule_search_locations=[str(ROOT / '4d-cbct-mc_amd')])
m = importlib.util.module_from_spec(spec); sys.modules['pkg']=m; spec.loader.exec_module(m)
from pkg.geometry import MCThoraxLikeGeometry, MCCIRSPhantomGeometry
from scipy import ndimage
which = sys.argv[1]
if which=='thorax':
    g = MCThoraxLikeGeometry(); B=16
else:
    g = MCCIRSPhantomGeometry.from_base_geometry(); B=8
mats, dens = g.materials, g.densities
print(mats.shape, dens.dtype, np.unique(mats))
# mu/rho at 60 keV by material number: crude (bone-like by density)
murho = np.where(dens>1.25, 0.30, np.where(dens>1.12, 0.25, 0.205)).astype(np.float32)
mu = (murho*dens/10.0).astype(np.float32)  # per mm
print('mu max', mu.max(), 'classes', np.unique(np.round(mu,4))[:20])
S = np.array(mats.shape)
nb = S//B
mub = mu[:nb[0]*B,:nb[1]*B,:nb[2]*B].reshape(nb[0],B,nb[1],B,nb[2],B).max(axis=(1,3,5))
Mg = mu.max()
air = mub < 0.001
# object box
idx = np.argwhere(~air); lo = idx.min(0)*B; hi = (idx.max(0)+1)*B
print('object box', lo, hi)
def radius_field(thr):
    dense = mub > thr
    if not dense.any(): return np.full(mub.shape, 1e9)
    d = ndimage.distance_transform_edt(~dense)  # in bricks, centre to centre
    return np.maximum(0, (d - 1.7321))*B  # mm: conservative (any point in brick to any point of a dense brick)
classes = [float(x) for x in sys.argv[2].split(',')]   # thresholds on mu (per mm) as fraction of Mg
rmin = float(sys.argv[3])
fields = [(t*Mg, radius_field(t*Mg)) for t in classes]
rng = np.random.default_rng(1)
N = 20000
def run(scheme):
    # start: isotropic-ish fan from a source at (−1000 mm x) through the box centre region
    c = (lo+hi)/2.0
    src = np.array([c[0]-1000.0, c[1], c[2]])
    tgt = np.stack([np.full(N,c[0]), rng.uniform(lo[1],hi[1],N), rng.uniform(lo[2],hi[2],N)],1)
    d = tgt-src; d/= np.linalg.norm(d,axis=1)[:,None]
    # enter object box
    t0 = (lo[0]-src[0])/d[:,0]
    p = src + d*(t0[:,None]+1e-3)
    alive = np.all((p>=lo)&(p<hi),axis=1)
    steps = np.zeros(N); caps=np.zeros(N); real=np.zeros(N)
    it=0
    while alive.any() and it<2000:
        it+=1
        ii = np.nonzero(alive)[0]
        pp = p[ii]; vi = pp.astype(int); bi = np.minimum(vi//B, nb-1)
        M = np.full(len(ii), Mg); R = np.full(len(ii), 1e9)
        if scheme:
            # pick the lowest class whose radius >= rmin
            chosen = np.zeros(len(ii),bool)
            for thr, f in fields:
                r = f[bi[:,0],bi[:,1],bi[:,2]]
                ok = (~chosen)&(r>=rmin)
                M[ok]=thr; R[ok]=r[ok]; chosen|=ok
        s = -np.log(rng.random(len(ii)))/M
        capped = s>R
        s = np.minimum(s,R)
        pp = pp + d[ii]*s[:,None]
        p[ii]=pp
        steps[ii]+=1; caps[ii]+=capped
        inside = np.all((pp>=lo)&(pp<hi),axis=1)
        alive[ii[~inside]] = False
        k = np.nonzero(inside & ~capped)[0]
        if len(k):
            v = pp[k].astype(int)
            muv = mu[v[:,0],v[:,1],v[:,2]]
            assert np.all(muv <= M[k]*1.0001+1e-9), (muv.max(), )
            hit = rng.random(len(k)) < muv/M[k]
            h = ii[k[hit]]
            real[h]+=1
            absorb = rng.random(len(h))<0.15
            alive[h[absorb]]=False
            hs = h[~absorb]
            u = rng.normal(size=(len(hs),3)); u/=np.linalg.norm(u,axis=1)[:,None]
            d[hs]=u
    return steps.mean(), caps.mean(), real.mean()
print('global   steps/caps/real per history', run(False))
print('scheme   steps/caps/real per history', run(True))

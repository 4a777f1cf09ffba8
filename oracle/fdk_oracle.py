"""CPU restatement (numpy) of circular cone-beam FDK reconstruction as `rtkfdk` performs it for the reference
(cbctmc/reconstruction/reconstruction.py:22-69: `rtkfdk --geometry geometry.xml --pad 1.0 --hann 1.0 --hannY 1.0
--dimension 464,250,464 --spacing 1 [--wpc ...] --short 360`).  TEST INFRASTRUCTURE: only tests/ may import it.

PARITY UNPINNED: RTK (github.com/RTKConsortium/RTK, used by the reference through the un-vendored `itk-rtk` wheel and a
docker image, docker/compile.sh:3-18) is not in /root/reference and not installable here, so this file restates the
published algorithm -- Feldkamp, Davis, Kress, JOSA A 1 (1984); the displaced-detector weights of Wang, Med. Phys. 29 (2002),
which RTK applies automatically to off-centre detectors; RTK's geometry conventions as documented in its
ThreeDCircularProjectionGeometry (Rit et al., J. Phys. Conf. Ser. 489 (2014)) -- and is pinned by analytic phantoms only
(tests/test_fdk.py), not by RTK output.  `--pad` (RTK: FFTRampImageFilter::TruncationCorrection, the heuristic of Ohnesorge et
al., Med. Phys. 27 (2000)) is restated from RTK's published implementation (rtkFFTProjectionsConvolutionImageFilter.hxx,
PadInputImageRegion): before the ramp every row is extended on both sides by next = ceil(pad x width) columns with the
point reflection of the data about the border value, 2 p(border) - p(mirror), feathered to zero by sin^0.75 -- see
`truncation_extension`; angular weights follow RTK's GetAngularGaps rule.

Geometry (RTK): fixed IEC frame, rotation axis y.  For gantry angle t: rotated coordinates p' = Ry(-t) p =
(x cos t - z sin t, y, x sin t + z cos t); source at p' = (0, 0, SID); detector plane z' = SID - SDD; a point projects to
u = SDD x' / (SID - z') - offset_x,  v = SDD y' / (SID - z') - offset_y  in the physical coordinates of the stack
(pixel (i, j) at (u0 + i du, v0 + j dv))."""
import numpy as np


def ramp_kernel(n_half: int, hann: float) -> np.ndarray:
    """Unit-spacing band-limited ramp h[-n_half..n_half] (h[0] = 1/4, h[odd] = -1/(pi n)^2), optionally apodised by a Hann
    window with cut-off hann * Nyquist (frequency response |f| * 0.5 (1 + cos(pi f / fc)) for |f| < fc, else 0)."""
    n = np.arange(-n_half, n_half + 1)
    h = np.zeros(n.shape, dtype=np.float64)
    h[n == 0] = 0.25
    odd = (n % 2) != 0
    h[odd] = -1.0 / (np.pi * n[odd]) ** 2
    if hann > 0.0:
        # apply the window in the frequency domain of a long zero-padded transform, come back, keep the same support
        m = 1
        while m < 8 * (2 * n_half + 1):
            m *= 2
        buf = np.zeros(m)
        buf[: n_half + 1] = h[n_half:]
        buf[-n_half:] = h[:n_half]
        f = np.fft.fftfreq(m)  # cycles per sample, Nyquist = 0.5
        fc = 0.5 * hann
        win = np.where(np.abs(f) < fc, 0.5 * (1.0 + np.cos(np.pi * f / fc)), 0.0)
        buf = np.real(np.fft.ifft(np.fft.fft(buf) * win))
        h = np.concatenate([buf[-n_half:], buf[: n_half + 1]])
    return h


def hann_y_kernel(hann_y: float, n_half: int = 8) -> np.ndarray:
    """Vertical low-pass of rtkfdk --hannY: Hann window with cut-off hann_y * Nyquist; [1/4, 1/2, 1/4] for 1.0."""
    if hann_y <= 0.0:
        return np.array([1.0])
    if hann_y == 1.0:
        return np.array([0.25, 0.5, 0.25])
    m = 4096
    f = np.fft.fftfreq(m)
    fc = 0.5 * hann_y
    win = np.where(np.abs(f) < fc, 0.5 * (1.0 + np.cos(np.pi * f / fc)), 0.0)
    k = np.real(np.fft.ifft(win))
    k = np.concatenate([k[-n_half:], k[: n_half + 1]])
    return k / k.sum()  # truncated support: keep the DC gain at exactly 1


def displaced_weights(u: np.ndarray, sdd: float) -> np.ndarray:
    """Weights for an off-centre detector over a full scan (Wang 2002): u = physical lateral coordinates of the columns
    relative to the central ray.  theta = the shorter extent; inside |u| <= theta conjugate rays share the weight smoothly
    (w(u) + w(-u) = 1), outside the ray is measured once (weight 1).  A centred detector gets 1/2 everywhere."""
    lo, hi = float(u.min()), float(u.max())
    if lo >= 0.0 or hi <= 0.0:
        return np.ones_like(u)  # the central ray is not on the detector: no conjugate overlap
    theta = min(-lo, hi)
    if abs((-lo) - hi) < 1e-9 * max(-lo, hi):
        return np.full_like(u, 0.5)
    sign = 1.0 if hi > -lo else -1.0  # long side
    s = sign * u
    w = np.where(s > theta, 1.0, 0.0)
    inside = np.abs(s) <= theta
    w = np.where(inside, 0.5 * (np.sin(np.pi * np.arctan(s / sdd) / (2.0 * np.arctan(theta / sdd))) + 1.0), w)
    return w


def symmetric_padding(nu, du, u0, off_min, off_max):
    """(columns to add on the left, on the right) so that the detector covers [-L, L] about the central ray for every
    projection offset in [off_min, off_max]; (0, 0) for a centred detector or when the central ray misses the detector."""
    lo, hi = u0 + off_min, u0 + (nu - 1) * du + off_max
    if lo >= 0.0 or hi <= 0.0:
        return 0, 0
    extent = max(-(u0 + off_min), -(u0 + off_max), u0 + (nu - 1) * du + off_min, u0 + (nu - 1) * du + off_max)
    pad_l = max(0, int(np.ceil((extent + (u0 + off_min)) / du - 1e-9)))
    pad_r = max(0, int(np.ceil((extent - (u0 + (nu - 1) * du + off_max)) / du - 1e-9)))
    return pad_l, pad_r


def truncation_extension(rows, pad):
    """rtkfdk --pad / TruncationCorrection: rows [..., n] -> [..., n + 2 next], next = min(ceil(pad * n), n - 1) columns
    added on each side.  Column at distance d (1..next) beyond a border holds w[d] * (2 p(border) - p(border -/+ d)) with
    w[d] = sin((next - d) pi / (2 next - 2)) ** 0.75 (1 at the border, 0 at the far end): the row continues with its own
    slope instead of dropping to zero, which is what makes the ramp ring at a truncated edge.  Returns (rows, next)."""
    rows = np.asarray(rows, dtype=np.float64)
    n = rows.shape[-1]
    nxt = int(min(np.ceil(pad * n), n - 1)) if pad > 0 else 0
    if nxt <= 0:
        return rows, 0
    d = np.arange(1, nxt + 1)
    w = np.sin((nxt - d) * np.pi / (2.0 * nxt - 2.0)) ** 0.75 if nxt > 1 else np.zeros(1)
    left = w * (2.0 * rows[..., :1] - rows[..., d])            # distance d left of column 0
    right = w * (2.0 * rows[..., -1:] - rows[..., n - 1 - d])  # distance d right of column n - 1
    return np.concatenate([left[..., ::-1], rows, right], axis=-1), nxt


def angular_gaps(gantry_deg):
    """Angular weight of every projection [rad]: half the distance to its two neighbours on the circle (the rule of RTK's
    ThreeDCircularProjectionGeometry::GetAngularGaps, which rtkfdk's FDKWeightProjectionFilter uses); projections at the
    same angle share their gap.  2 pi / n for a uniform full arc."""
    a = np.mod(np.asarray(gantry_deg, dtype=np.float64), 360.0)
    uniq, inverse, counts = np.unique(np.round(a, 9), return_inverse=True, return_counts=True)
    m = uniq.size
    if m == 1:
        gaps = np.array([360.0])
    elif m == 2:
        gaps = np.array([180.0, 180.0])
    else:
        prev, nxt = np.roll(uniq, 1), np.roll(uniq, -1)
        gaps = 0.5 * np.mod(nxt - prev, 360.0)
    return np.deg2rad(gaps[inverse] / counts[inverse])


def reconstruct(proj, du, dv, u0, v0, sid, sdd, gantry_deg, off_x, off_y, dim, spacing, origin=None, hann=0.0, hann_y=0.0, wpc=None, pad=0.0):
    """proj [n][nv][nu] line integrals -> volume [nz][ny][nx] (float64).  origin = centre of voxel (0,0,0); None = centred.
    pad: rtkfdk --pad (truncation correction, see truncation_extension); 0 = rows are zero-padded only."""
    proj = np.asarray(proj, dtype=np.float64)
    n, nv, nu = proj.shape
    nx, ny, nz = dim
    sx, sy, sz = spacing
    if origin is None:
        origin = (-(nx - 1) / 2 * sx, -(ny - 1) / 2 * sy, -(nz - 1) / 2 * sz)
    if wpc is not None and len(wpc):
        acc = np.zeros_like(proj)
        pw = np.ones_like(proj)
        for c in wpc:
            acc += c * pw
            pw *= proj
        proj = acc
    off_x = np.broadcast_to(np.asarray(off_x, dtype=np.float64), (n,))
    off_y = np.broadcast_to(np.asarray(off_y, dtype=np.float64), (n,))
    # An off-centre detector is first padded with zero columns on its short side until it is symmetric about the central
    # ray (RTK: DisplacedDetectorImageFilter "weighs and pads"): the ramp-filtered, weighted rows are non-zero beyond the
    # physical edge, and the conjugate views need exactly those values.
    pad_l, pad_r = symmetric_padding(nu, du, u0, float(off_x.min()), float(off_x.max()))
    nu_p, u0_p = nu + pad_l + pad_r, u0 - pad_l * du
    nxt = int(min(np.ceil(pad * nu_p), nu_p - 1)) if pad > 0 else 0
    nu_e = nu_p + 2 * nxt  # row length the ramp sees
    h = ramp_kernel(nu_e - 1, hann)
    ky = hann_y_kernel(hann_y)
    vol = np.zeros((nz, ny, nx), dtype=np.float64)
    X = origin[0] + sx * np.arange(nx)
    Y = origin[1] + sy * np.arange(ny)
    Z = origin[2] + sz * np.arange(nz)
    dbeta = angular_gaps(gantry_deg)  # [rad] per projection
    for k in range(n):
        # physical coordinates of the pixel centres relative to the central ray
        up = u0 + du * np.arange(nu) + off_x[k]
        vp = v0 + dv * np.arange(nv) + off_y[k]
        w_cos = sdd / np.sqrt(sdd * sdd + up[None, :] ** 2 + vp[:, None] ** 2)
        w_dis = displaced_weights(up, sdd)
        p = np.pad(proj[k] * w_cos * w_dis[None, :], ((0, 0), (pad_l, pad_r)))
        p, _ = truncation_extension(p, pad)
        # ramp along u (linear convolution, zero padded), scaled to the real detector: 1/du * SDD/SID; only the columns of the
        # (symmetrically padded) detector are kept
        q = np.stack([np.convolve(row, h, mode="full")[nu_e - 1 + nxt: nu_e - 1 + nxt + nu_p] for row in p])
        q *= (sdd / sid) / du
        if ky.size > 1:
            hk = ky.size // 2
            qp = np.pad(q, ((hk, hk), (0, 0)), mode="edge")
            q = sum(ky[j] * qp[j: j + nv] for j in range(ky.size))
        t = np.deg2rad(gantry_deg[k])
        c, s = np.cos(t), np.sin(t)
        xr = X[None, :] * c - Z[:, None] * s          # [nz][nx]
        zr = X[None, :] * s + Z[:, None] * c
        U = sid - zr                                   # distance from the source along the central axis
        mag = sdd / U
        fu = (mag * xr - off_x[k] - u0_p) / du         # fractional column index of the padded rows  [nz][nx]
        wgt = dbeta[k] * (sid / U) ** 2
        iu = np.floor(fu).astype(np.int64)
        au = fu - iu
        ok_u = (iu >= 0) & (iu < nu_p - 1)
        iu_c = np.clip(iu, 0, nu_p - 2)
        for j in range(ny):
            fv = (mag * Y[j] - off_y[k] - v0) / dv
            iv = np.floor(fv).astype(np.int64)
            av = fv - iv
            ok = ok_u & (iv >= 0) & (iv < nv - 1)
            iv_c = np.clip(iv, 0, nv - 2)
            val = ((1 - av) * ((1 - au) * q[iv_c, iu_c] + au * q[iv_c, iu_c + 1]) + av * ((1 - au) * q[iv_c + 1, iu_c] + au * q[iv_c + 1, iu_c + 1]))
            vol[:, j, :] += np.where(ok, wgt * val, 0.0)
    return vol


def sphere_projections(mu, radius, centre, n, nu, nv, du, dv, u0, v0, sid, sdd, gantry_deg, off_x, off_y):
    """Exact line integrals of a uniform sphere (attenuation mu, given centre in the fixed frame) under the geometry above."""
    out = np.zeros((n, nv, nu), dtype=np.float64)
    cx, cy, cz = centre
    for k in range(n):
        t = np.deg2rad(gantry_deg[k])
        c, s = np.cos(t), np.sin(t)
        # work in the rotated frame: source (0,0,sid), sphere centre rotated
        pc = np.array([cx * c - cz * s, cy, cx * s + cz * c])
        up = u0 + du * np.arange(nu) + (off_x[k] if np.ndim(off_x) else off_x)
        vp = v0 + dv * np.arange(nv) + (off_y[k] if np.ndim(off_y) else off_y)
        dx, dy = np.meshgrid(up, vp)                    # detector point (dx, dy, sid - sdd)
        d = np.stack([dx, dy, np.full_like(dx, -sdd)], axis=-1)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        oc = np.array([0.0, 0.0, sid]) - pc
        b = d @ oc
        disc = b * b - (oc @ oc - radius * radius)
        out[k] = np.where(disc > 0, 2.0 * mu * np.sqrt(np.maximum(disc, 0.0)), 0.0)
    return out

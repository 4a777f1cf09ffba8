"""`cpu_baseline`: the CPU oracle (the reference's algorithm restated in C) timed on the GPU box's host cores on a bounded sample of the same workload."""
from __future__ import annotations

import os
import time

import numpy as np

from .common import usable_cpus

def cpu_baseline(ctx, seconds_budget: float = 16.0):
    """The restated CPU oracle (oracle/mcgpu_oracle.c, LIBM math == reference arithmetic; the loop being timed is the
    reference's MC-GPU_v1.3.cu:913-958) on a bounded sample of the same workload, on this host's cores.  Reported, never the
    target.  Also returns the sample's image and its sum of squared weights (for `check`)."""
    from .common import checker_paths
    checker_paths()
    import oracle_lib as ol
    import parity
    T = parity.tables_from_context(ctx)
    usable = usable_cpus()
    hpt = 150
    image = np.zeros(T.image_size(), dtype=np.uint64)
    w2 = np.zeros(T.image_size(), dtype=np.uint64)
    cnt = ol.OracleCounters()
    batch0 = [0]

    def timed(threads, target_s):
        """Rate with `threads` OpenMP threads: chunks sized from the previous one until `target_s` of timed work."""
        nb, done, elapsed = max(threads * 16, 64), 0, 0.0
        while elapsed < target_s:
            t0 = time.perf_counter()
            T.track(0, 42, batch0[0], nb, hpt, ol.MATH_LIBM, n_threads=threads, image=image, counters=cnt, w2=w2)
            dt = time.perf_counter() - t0
            batch0[0] += nb
            if dt > 0.3 or threads == 1:  # chunks too short to time OpenMP start-up fairly are warm-up only
                done += nb
                elapsed += dt
            nb = int(max(threads * 16, min(nb * 1.5 / max(dt, 1e-3), 4e6)))
        return done * hpt / elapsed, done * hpt, elapsed

    curve = {}
    points = sorted({t for t in (1, 4, 16, 64, usable) if t <= usable})
    share = seconds_budget / (len(points) + 1)
    for t in points:
        rate, n, secs = timed(t, share * (2.0 if t == usable else 1.0))
        curve[str(t)] = rate
    c = cnt.as_dict()
    h = float(c["histories"])
    per_hist = {k: round(c[k] / h, 4) for k in ("steps", "voxel_reads", "mfp_reads", "woodcock_reads", "compton", "rayleigh", "photo", "rng", "tally_calls", "tally_hits")}
    algo = 8 * per_hist["voxel_reads"] + 24 * per_hist["mfp_reads"] + 8 * per_hist["woodcock_reads"] + 16 * per_hist["tally_calls"]
    out = {
        "value": curve[str(usable)], "unit": "histories/s", "cores": usable, "kind": "port",
        "sample": f"{n} histories of projection 0 of the same workload in {secs:.1f} s on {usable} threads (of {int(h)} in the whole thread curve), "
                  "OpenMP over RANECU batches (oracle/mcgpu_oracle.c, libm math)",
        "per_core_value": curve["1"], "threads_curve_histories_per_s": curve,
        "host": {"os_cpu_count": os.cpu_count(), "sched_affinity": len(os.sched_getaffinity(0)), "usable": usable},
        "events_per_history": per_hist, "algorithmic_bytes_per_history_from_these_counts": round(algo, 1),
    }
    return out, image, w2.astype(np.float64) * (1024.0 ** 2), int(h)

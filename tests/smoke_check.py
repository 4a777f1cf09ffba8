"""__graft_entry__.smoke(): one small projection on cuda:0 through the C ABI, checked against the oracle."""
from __future__ import annotations

import tempfile

import numpy as np


def run():
    import cases
    import oracle_lib as ol
    import parity
    eng = cases.pkg.engine
    with tempfile.TemporaryDirectory() as tmp:
        inp = cases.build_case("catphan64", tmp)
        with eng.create(inp, device=0) as ctx:
            T = parity.tables_from_context(ctx)
            nb, hpt = 512, 50
            img_gpu, secs, done = ctx.run_projection(0, nb, mode="compat", seed=42, hpt=hpt)
            img_cpu, _ = T.track(0, 42, 0, nb, hpt, ol.MATH_PORTABLE, n_threads=4)
            same = np.array_equal(img_gpu.reshape(-1), img_cpu)
            print(f"smoke: compat kernel {done} histories in {secs*1e3:.2f} ms; bit-identical to CPU oracle: {same}; "
                  f"sum={int(img_gpu.sum())}")
            if not same:
                raise SystemExit("smoke FAILED: GPU compat tallies differ from the CPU oracle")
            img_fast, secs, done = ctx.run_projection(0, 2_000_000, mode="fast", seed=42)
            frac = img_fast.sum() / done / (img_cpu.sum() / (nb * hpt))
            print(f"smoke: fast kernel {done} histories in {secs*1e3:.2f} ms ({done/secs:.3e} hist/s); detected-energy ratio fast/oracle = {frac:.4f}")
            if not (0.97 < frac < 1.03):
                raise SystemExit("smoke FAILED: fast kernel detected energy off by more than 3%")


if __name__ == "__main__":
    run()

#!/usr/bin/env python3
"""Write gpurun_out/fast_pin.json (GPU): SHA-256 of the FAST kernel's integer tallies on small cases.  Copy the file to
tests/golden/fast_pin.json after a DELIBERATE change of the FAST arithmetic or random-number use; the test
test_fast_kernel_tallies_are_pinned then fails whenever a build changes a single tally word by accident (compiler, flags)."""
import hashlib, json, sys
from pathlib import Path
import tempfile
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
PIN_CASES = [("catphan64_ct", 1, 600_000, 11), ("tissue22", 0, 400_000, 12), ("cirs76", 2, 400_000, 13), ("air", 0, 200_000, 14)]


def compute(mode="fast"):
    eng = cases.pkg.engine
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, p, n, seed in PIN_CASES:
            with eng.create(cases.build_case(name, Path(tmp) / name), device=0) as ctx:
                img, _, done = ctx.run_projection(p, n, mode=mode, seed=seed)
                out[name] = {"projection": p, "histories": n, "seed": seed, "sum": int(img.sum()), "sha256": hashlib.sha256(img.tobytes()).hexdigest()}
    return out


if __name__ == "__main__":
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    for mode in ("fast", "fast64"):
        (ROOT / "gpurun_out" / f"{mode}_pin.json").write_text(json.dumps(compute(mode), indent=1))
        print(mode, json.dumps(compute(mode), indent=1))

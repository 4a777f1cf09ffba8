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

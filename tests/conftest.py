import os
import sys
from pathlib import Path

import pytest

try:  # torch (with the ROCm libraries it bundles) must be loaded BEFORE the engine library brings up HIP: loaded after it,
    import torch  # noqa: F401  -- torch finds no GPU ("No HIP GPUs are available"); some GPU tests hand torch tensors to the engine
except Exception:  # noqa: BLE001
    torch = None

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def case_dir(tmp_path_factory):
    """Builds the small test cases (geometry + .in files) once per session."""
    import cases
    base = tmp_path_factory.mktemp("mcgpu_cases")
    built = {}

    def get(name, **overrides):
        key = (name, tuple(sorted(overrides.items())))
        if key not in built:
            sub = base / (name + ("_" + str(len(built)) if overrides else ""))
            built[key] = cases.build_case(name, sub, **overrides)
        return built[key]

    return get


@pytest.fixture(scope="session")
def engine():
    import cases
    eng = cases.pkg.engine
    eng.load_library()
    return eng

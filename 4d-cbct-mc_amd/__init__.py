"""MI355X-native Monte Carlo CBCT projection engine behind the `cbctmc.mc` file contract.

The directory name (`4d-cbct-mc_amd`) is not a Python identifier; import it through
`__graft_entry__.load_package()` (registers the package as `cbctmc_amd`).

Layout: `csrc/` holds the HIP kernels, the C++ host model and the C ABI (`include/mcgpu_amd.h`);
the Python modules mirror the reference's host-side interface for this path
(`cbctmc/mc/simulation.py`, `geometry.py`, `materials.py`, `defaults.py`, `projection.py`).
"""
from . import defaults, geometry, materials, reconstruction, respiratory, sharding, simulation, workloads  # noqa: F401

__all__ = ["defaults", "geometry", "materials", "reconstruction", "respiratory", "sharding", "simulation", "workloads", "engine"]


def __getattr__(name):
    if name == "engine":
        import importlib
        return importlib.import_module(__name__ + ".engine")
    raise AttributeError(name)

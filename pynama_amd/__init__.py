"""pynama_amd -- MI355X-native implementation of Pynama's finite/spectral-element hot path.

Python host code keeps the reference's class / method names (``Spectral``, ``DMPlexDom``,
``Mat``, ``KspSolver``, ``FreeSlip`` ...) and calls hand-written HIP kernels through the C ABI
of ``include/pynama_hip.h`` (ctypes).  No PyTorch, no CPU fallback.

The reference imports its modules with ``src/`` as the root (``from elements.spectral import
Spectral``).  ``install_reference_layout()`` registers the same top-level names as aliases of
this package's sub-packages, so reference-style scripts and tests run unchanged.
"""
import importlib
import sys

__version__ = "0.1.0"

_SUBPACKAGES = {
    "elements": ("element", "utilities", "spectral", "simplex"),
    "domain": ("dmplex", "gmsh"),
    "viewer": ("xml_generator", "paraviewer", "hdf5_writer"),
    "matrices": ("mat_generator", "mat_ns"),
    "solver": ("ksp_solver",),
    "common": ("timer", "nswalls", "options", "comm"),
    "cases": ("base_problem", "uniform", "custom_func", "cavity"),
}


def install_reference_layout():
    """Alias ``elements``, ``domain``, ``matrices``, ``solver``, ``cases``, ``common`` (and their
    modules) to the pynama_amd implementations, one module object each."""
    for pkg, mods in _SUBPACKAGES.items():
        p = importlib.import_module(f"pynama_amd.{pkg}")
        sys.modules.setdefault(pkg, p)
        for m in mods:
            try:
                sub = importlib.import_module(f"pynama_amd.{pkg}.{m}")
            except ModuleNotFoundError:
                continue
            sys.modules.setdefault(f"{pkg}.{m}", sub)
    # the reference keeps IndicesManager in domain/indices.py; here it lives next to the lattice numbering it serves
    sys.modules.setdefault("domain.indices", sys.modules["domain.dmplex"])

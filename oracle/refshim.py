"""Import shim for running the *unmodified* upstream reference in the authoring container.

TEST INFRASTRUCTURE ONLY.  Used exclusively by tests/golden/make_golden.py (which writes the
committed .npz fixtures) — never by the product path, never on the GPU box (the reference does
not travel).  Nothing from the reference is copied: this file only patches the *environment*
so that `import world` from /root/reference works under NumPy 2.x / SciPy 1.15 without numba
(SURVEY.md §8(c)).
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def install(reference_root: str = REFERENCE_ROOT):
    import numpy as np
    import scipy.signal
    import scipy.signal.windows

    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")

        class _Sig:
            def __getitem__(self, item):
                return self

            def __call__(self, *a, **k):
                return self

        def jit(*sig, **kw):
            if len(sig) == 1 and callable(sig[0]) and not isinstance(sig[0], (tuple, _Sig)):
                return sig[0]

            def deco(fn):
                return fn

            return deco

        nb.jit = jit
        nb.njit = jit
        nb.float64 = _Sig()
        nb.int64 = _Sig()
        sys.modules["numba"] = nb
    import numpy.matlib  # noqa: F401 — world/swipe.py uses np.matlib without importing it
    if not hasattr(np, "int"):
        np.int = int  # removed alias used by the reference
    if not hasattr(scipy.signal, "hanning"):
        scipy.signal.hanning = scipy.signal.windows.hann
    try:
        import matplotlib

        matplotlib.use("Agg")
    except Exception:
        pass
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


def load():
    """Return the reference's `world` package modules as a namespace."""
    install()
    import importlib

    mods = {}
    for name in ("dio", "stonemask", "harvest", "cheaptrick", "d4c", "d4cRequiem", "synthesis",
                 "synthesisRequiem", "get_seeds_signals", "swipe", "main"):
        mods[name] = importlib.import_module("world." + name)
    return types.SimpleNamespace(**mods)

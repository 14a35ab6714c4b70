"""SWIPE' is outside the accelerated hot path.  The symbol exists so
that `world.main` keeps the reference's import surface (world/main.py:23)."""


def swipe(fs, x, plim, dt=0.005, sTHR=0.3):
    raise NotImplementedError("f0_method='swipe' is not part of the MI355X build (use 'dio' or 'harvest')")

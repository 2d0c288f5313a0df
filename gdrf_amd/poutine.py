"""``pyro.poutine.scale`` as the reference uses it (gdrf/train_script.py:365-369):
``scale = poutine.scale(scale=1/len(xs)); SVI(model=scale(model.model), guide=scale(model.guide), ...)``.
Every site's log-probability is multiplied by ``scale``; here that is the 1/N_global factor the
HIP epilogue applies (gdrf_step_finish).
"""
from __future__ import annotations


class ScaledFn:
    def __init__(self, fn, scale: float):
        self.fn = fn
        self.scale = float(scale)
        self.__self__ = getattr(fn, "__self__", None)

    def __call__(self, *a, **k):
        return self.fn(*a, **k)


class scale:  # noqa: N801  (name mirrors pyro.poutine.scale)
    def __init__(self, fn=None, scale: float = 1.0):
        if not scale > 0:
            raise ValueError("scale must be positive")
        self._scale = float(scale)
        self._fn = fn

    def __call__(self, fn):
        return ScaledFn(fn, self._scale)

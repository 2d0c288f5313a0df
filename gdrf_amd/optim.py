"""``pyro.optim.{Adam,AdamW,ClippedAdam}({"lr": lr})`` as built at gdrf/train_script.py:73-87,325-327.

Pyro keeps one torch optimizer per unconstrained parameter tensor, all stepping together; the
HIP path applies the same element-wise update to the flat parameter vector in one launch
(gdrf_adam; arithmetic in SURVEY.md A.5).  ``get_state``/``set_state`` (train_script.py:348,495)
expose per-parameter ``{"step", "exp_avg", "exp_avg_sq"}`` like torch's optimizer state.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


class PyroOptimLike:
    mode = "adam"
    defaults: Dict[str, object] = {}

    def __init__(self, optim_args: Optional[dict] = None, clip_args=None):
        if callable(optim_args):
            raise NotImplementedError("per-parameter optim_args callables")
        if clip_args:
            raise NotImplementedError("clip_args")
        args = dict(self.defaults)
        args.update(optim_args or {})
        unknown = set(args) - set(self.defaults)
        if unknown:
            raise ValueError(f"unsupported optimizer arguments: {sorted(unknown)}")
        self.args = args
        self.lr = float(args["lr"])
        self._engine = None
        self._pending_state = None

    # -- bound by SVI at its first step
    def _bind(self, engine):
        if self._engine is engine:
            return
        self._engine = engine
        if self._pending_state is not None:
            self.set_state(self._pending_state)
            self._pending_state = None

    def _step(self):
        e = self._engine
        a = self.args
        if self.mode == "clippedadam":
            self.lr *= float(a["lrd"])         # pyro's ClippedAdam decays lr BEFORE it forms step_size
        e.adam(self.mode, self.lr, betas=tuple(a["betas"]), eps=float(a["eps"]),
               weight_decay=float(a.get("weight_decay", 0.0)), clip=float(a.get("clip_norm", 10.0)))

    def get_state(self) -> dict:
        e = self._engine
        if e is None:
            return self._pending_state or {}
        m, v = e.named_views(e.exp_avg), e.named_views(e.exp_avg_sq)
        return {name: {"step": e.opt_step, "exp_avg": m[name].detach().clone(), "exp_avg_sq": v[name].detach().clone(),
                       "lr": self.lr} for name in e.param_names}

    def set_state(self, state: dict):
        e = self._engine
        if e is None:
            self._pending_state = state
            return
        m, v = e.named_views(e.exp_avg), e.named_views(e.exp_avg_sq)
        for name, st in state.items():
            if name not in m:
                continue
            m[name].copy_(torch.as_tensor(st["exp_avg"]).to(m[name]))
            v[name].copy_(torch.as_tensor(st["exp_avg_sq"]).to(v[name]))
            e.opt_step = int(st["step"])
            self.lr = float(st.get("lr", self.lr))


class Adam(PyroOptimLike):
    mode = "adam"
    defaults = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8)


class AdamW(PyroOptimLike):
    mode = "adamw"
    defaults = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)


class ClippedAdam(PyroOptimLike):
    mode = "clippedadam"
    defaults = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clip_norm=10.0, lrd=1.0)


def _unsupported(name):
    class _U:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"pyro.optim.{name} is a plain registry entry of the reference "
                                      "(train_script.py:73-87) outside this build's hot path; Adam, AdamW and ClippedAdam are in")
    _U.__name__ = name
    return _U


OPTIMIZER_DICT = {"adam": Adam, "adamw": AdamW, "clippedadam": ClippedAdam}
for _n in ["AdagradRMSProp", "DCTAdam", "Adadelta", "Adagrad", "SparseAdam", "Adamax", "ASGD", "SGD", "Rprop", "RMSprop"]:
    OPTIMIZER_DICT[_n.lower()] = _unsupported(_n)

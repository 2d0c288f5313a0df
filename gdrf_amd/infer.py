"""``pyro.infer.SVI`` / ``Trace_ELBO`` / ``TraceGraph_ELBO`` as used by gdrf/train_script.py:88-92,
330-335,365-371,467:  ``svi.step(xs=, ws=, subsample=False) -> float`` (loss = -ELBO, every site
scaled by the poutine.scale factor).

For ``SparseMultinomialGDRF`` every guide site is reparameterised, so Trace_ELBO and
TraceGraph_ELBO give the same value and gradient; one HIP evaluation serves both.
"""
from __future__ import annotations

from typing import Optional

import torch

from .poutine import ScaledFn


class _ELBO:
    def __init__(self, num_particles: int = 1, max_plate_nesting: int = float("inf"), vectorize_particles: bool = False,
                 **kwargs):
        if int(num_particles) < 1:
            raise ValueError("num_particles must be >= 1")
        self.num_particles = int(num_particles)
        self.max_plate_nesting = max_plate_nesting
        self.vectorize_particles = vectorize_particles


class Trace_ELBO(_ELBO):  # noqa: N801
    pass


class TraceGraph_ELBO(_ELBO):  # noqa: N801
    pass


class RenyiELBO(_ELBO):
    """pyro.infer.RenyiELBO(alpha=0, num_particles=2, ...) (gdrf/train_script.py:88-92,330-335): the Renyi alpha-divergence
    bound  -(logsumexp_p((1 - alpha) elbo_p) - log P) / (1 - alpha); alpha = 0 is the importance-weighted (IWAE) bound.
    The particles' payloads are combined with their normalised importance weights on the device (Engine._local_and_finish)."""

    def __init__(self, alpha: float = 0.0, num_particles: int = 2, max_plate_nesting: int = float("inf"),
                 vectorize_particles: bool = False, **kwargs):
        if float(alpha) == 1.0:
            raise ValueError("The order alpha should not be equal to 1. Please use Trace_ELBO class for the case alpha = 1.")
        super().__init__(num_particles=num_particles, max_plate_nesting=max_plate_nesting, vectorize_particles=vectorize_particles)
        self.alpha = float(alpha)


OBJECTIVE_DICT = {"elbo": Trace_ELBO, "graphelbo": TraceGraph_ELBO, "renyielbo": RenyiELBO}


def _unwrap(fn):
    scale = 1.0
    while isinstance(fn, ScaledFn):
        scale *= fn.scale
        fn = fn.fn
    owner = getattr(fn, "__self__", None)
    if owner is None or not hasattr(owner, "_engine_for"):
        raise TypeError("SVI needs model=<gdrf_amd model>.model and guide=<the same model>.guide "
                        "(optionally wrapped in gdrf_amd.poutine.scale)")
    return owner, scale


class SVI:
    def __init__(self, model, guide, optim, loss, **kwargs):
        m_owner, m_scale = _unwrap(model)
        g_owner, g_scale = _unwrap(guide)
        if m_owner is not g_owner:
            raise ValueError("model and guide must be methods of the same gdrf_amd model")
        if abs(m_scale - g_scale) > 1e-15 * max(m_scale, g_scale):
            raise ValueError("model and guide carry different poutine.scale factors")
        if not isinstance(loss, _ELBO):
            raise TypeError("loss must be gdrf_amd.infer.Trace_ELBO, TraceGraph_ELBO or RenyiELBO")
        self.gdrf = m_owner
        self.scale = m_scale
        self.optim = optim
        self.loss = loss
        self.row_offset = 0            # global index of this rank's first observation (Philox key)
        self.steps_taken = 0

    def step(self, *args, xs=None, ws=None, subsample=False, eps: Optional[torch.Tensor] = None) -> float:
        """One SVI step on (xs, ws) (this rank's shard when torch.distributed is initialised).
        ``eps`` (K, n) injects the N(0,1) draw of the guide's mu site for seed-for-seed parity."""
        if args:
            xs = args[0]
            ws = args[1] if len(args) > 1 else ws
        model = self.gdrf
        xs_s, ws_d = model._prepare_inputs(xs, ws)
        eng = model._engine_for(xs_s.shape[0])
        self.optim._bind(eng)
        n = xs_s.shape[0]
        eps = self._eps(eng, eps, n)
        xs_g = model._guide_inputs(xs_s)
        eng.loss_and_grads(xs_s, ws_d, eps, n_global=1.0 / self.scale, renyi_alpha=getattr(self.loss, "alpha", None),
                           mean=model._mean_values(xs_s), xs_guide=xs_g, mean_guide=None if xs_g is None else model._mean_values(xs_g))
        self.optim._step()
        out = eng.read_out()
        self.steps_taken += 1
        self.last = out
        return float(out["loss"])

    def evaluate_loss(self, *args, xs=None, ws=None, subsample=False, eps=None) -> float:
        if args:
            xs = args[0]
            ws = args[1] if len(args) > 1 else ws
        model = self.gdrf
        xs_s, ws_d = model._prepare_inputs(xs, ws)
        eng = model._engine_for(xs_s.shape[0])
        n = xs_s.shape[0]
        xs_g = model._guide_inputs(xs_s)
        eng.loss_and_grads(xs_s, ws_d, self._eps(eng, eps, n), n_global=1.0 / self.scale,
                           renyi_alpha=getattr(self.loss, "alpha", None), mean=model._mean_values(xs_s), xs_guide=xs_g,
                           mean_guide=None if xs_g is None else model._mean_values(xs_g))
        return float(eng.read_out()["loss"])

    def _eps(self, eng, eps, n):
        """(P, K, n) standard normals: injected, or Philox keyed by (seed; global row, topic, step * P + particle)."""
        P = self.loss.num_particles
        if eps is not None:
            eps = torch.as_tensor(eps).to(device=eng.device, dtype=eng.dtype).contiguous()
            if eps.dim() == 2:
                eps = eps.unsqueeze(0)
            if eps.shape[0] != P:
                raise ValueError(f"eps carries {eps.shape[0]} particles, the objective has num_particles={P}")
            return eps
        out = torch.empty(P, eng.K, n, dtype=eng.dtype, device=eng.device)
        for p in range(P):
            eng.fill_eps(self.gdrf.rng_seed, self.steps_taken * P + p, self.row_offset, n, out=out[p])
        return out

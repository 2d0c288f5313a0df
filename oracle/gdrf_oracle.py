"""CPU oracle for the GDRF SVI ELBO hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product path (``gdrf_amd``) never does.

PARITY UNPINNED: the reference (san-soucie/gdrf v0.1.3) holds no golden vectors,
known-answer tests or fixtures for this path (``tests/test_gdrf.py:9-22`` is a
placeholder) and its arithmetic lives in pyro-ppl 1.8.0 (``poetry.lock:1169-1170``),
which is neither vendored under /root/reference nor installed here.  This file
restates that arithmetic on top of the torch pieces pyro
itself delegates to (``torch.distributions``, ``torch.linalg``, ``torch.optim``,
autograd) and is pinned only by the analytic known-answer cases of SURVEY A.6
(``tests/test_oracle.py``).

Two flavours:

* ``RefShapedGDRF`` -- "reference-shaped": same op sequence as the Pyro trace of
  ``SparseMultinomialGDRF`` (guide then model, ``conditional`` evaluated twice,
  ``W @ S_2D`` materialised, autograd backward, one torch optimizer per
  parameter).  This is the parity oracle and the timed CPU baseline.
* ``fused_elbo_and_grads`` -- "fused-shaped": one evaluation with the
  hand-derived backward in exactly the factorisation the HIP
  kernels use.  Used to validate that derivation against autograd.

Reference call sites restated here (paths under /root/reference):
  gdrf/models/sparse_gdrf.py:16-123   parameters and initialisation
  gdrf/models/sparse_gdrf.py:161-186  log_topic_probs
  gdrf/models/sparse_gdrf.py:323-373  SparseMultinomialGDRF.model
  gdrf/models/sparse_gdrf.py:375-409  SparseMultinomialGDRF.guide
  gdrf/models/abstract_gdrf.py:17-22,57-84,113-139
  gdrf/models/utils.py:6-40           validate_dirichlet_param, jittercholesky
  gdrf/train_script.py:251-273,365-371,467-472
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.distributions import Dirichlet, Multinomial, Normal, constraints, transform_to

SQRT5 = 5.0 ** 0.5


# --------------------------------------------------------------------------
# kernels (pyro.contrib.gp.kernels.isotropic, pyro-ppl 1.8.0)
# --------------------------------------------------------------------------
def square_scaled_dist(X: torch.Tensor, Z: torch.Tensor, lengthscale: torch.Tensor) -> torch.Tensor:
    """Isotropy._square_scaled_dist: expanded ||x||^2 - 2 x.z + ||z||^2, clamped at 0."""
    sX = X / lengthscale
    sZ = Z / lengthscale
    X2 = (sX ** 2).sum(1, keepdim=True)
    Z2 = (sZ ** 2).sum(1, keepdim=True)
    XZ = sX.matmul(sZ.t())
    r2 = X2 - 2 * XZ + Z2.t()
    return r2.clamp(min=0)


def kernel_matrix(kind: str, X: torch.Tensor, Z: torch.Tensor, lengthscale, variance, scale_mixture=None) -> torch.Tensor:
    r2 = square_scaled_dist(X, Z, lengthscale)
    if kind == "rbf":
        return variance * torch.exp(-0.5 * r2)
    if kind == "rationalquadratic":
        # pyro 1.8.0 kernels.isotropic.RationalQuadratic.forward (third-party, not in the tree; published formula):
        # variance * (1 + (0.5 / scale_mixture) * r2).pow(-scale_mixture), scale_mixture a positive PyroParam, default 1
        return variance * (1 + (0.5 / scale_mixture) * r2).pow(-scale_mixture)
    r = (r2 + 1e-12).sqrt()
    if kind == "matern52":
        s5r = SQRT5 * r
        return variance * (1 + s5r + (5.0 / 3.0) * r ** 2) * torch.exp(-s5r)
    if kind == "matern32":
        s3r = 3.0 ** 0.5 * r
        return variance * (1 + s3r) * torch.exp(-s3r)
    if kind == "exponential":
        return variance * torch.exp(-r)
    raise ValueError(f"unknown kernel {kind}")


def kernel_diag(X: torch.Tensor, variance: torch.Tensor) -> torch.Tensor:
    return variance.expand(X.size(0))


# --------------------------------------------------------------------------
# gdrf/models/utils.py:27-40  (cumulative, in-place jitter)
# --------------------------------------------------------------------------
def jittercholesky(Kff: torch.Tensor, N: int, jitter: float, maxjitter: int, noise: float = 0.0,
                   force_level: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """Returns (L, njitter).  ``force_level`` pins the number of failed attempts
    (used by parity tests so both sides add the same cumulative jitter)."""
    Kff = Kff + torch.eye(N, dtype=Kff.dtype) * noise
    njitter = 0
    eye = torch.eye(N, dtype=Kff.dtype)
    while njitter < maxjitter:
        Kff = Kff + eye * (jitter * (10 ** njitter))
        if force_level is not None and njitter < force_level:
            njitter += 1
            continue
        L, info = torch.linalg.cholesky_ex(Kff)
        if int(info) == 0 and bool(torch.isfinite(L).all()):
            return L, njitter
        njitter += 1
    raise RuntimeError("reached max jitter, covariance is unstable")


def jitter_total(jitter: float, level: int) -> float:
    """Total diagonal added after ``level`` failed attempts (attempt ``level`` succeeds)."""
    return sum(jitter * (10 ** n) for n in range(level + 1))


# --------------------------------------------------------------------------
# pyro.contrib.gp.util.conditional, full_cov=False, whiten=True  
# --------------------------------------------------------------------------
def conditional(kind, Xnew, Z, lengthscale, variance, u_loc, u_scale_tril, Lff, scale_mixture=None, whiten=True):
    M = Z.size(0)
    K = u_loc.size(0)
    N = Xnew.size(0)
    Kfs = kernel_matrix(kind, Z, Xnew, lengthscale, variance, scale_mixture)            # (M,N)
    v_2D = u_loc.reshape(-1, M).t()                                      # (M,K)
    S_2D = u_scale_tril.reshape(-1, M, M).permute(1, 2, 0).reshape(M, -1)  # (M, M*K): col = j*K + k
    W = torch.linalg.solve_triangular(Lff, Kfs, upper=False).t()         # (N,M)
    if not whiten:
        # pyro conditional, whiten=False branch: [f_loc, Kfs, f_scale_tril] are packed and solved against Lff together, i.e.
        # the mean and the scale factor are given in the unwhitened space: v = Lff^-1 f_loc, S = Lff^-1 f_scale_tril
        v_2D = torch.linalg.solve_triangular(Lff, v_2D, upper=False)
        S_2D = torch.linalg.solve_triangular(Lff, S_2D, upper=False)
    loc = W.matmul(v_2D).t().reshape(K, N)
    Kssdiag = kernel_diag(Xnew, variance)
    Qssdiag = W.pow(2).sum(dim=-1)
    var = (Kssdiag - Qssdiag).clamp(min=0)
    W_S = W.matmul(S_2D).reshape(N, M, K).permute(2, 0, 1)               # (K,N,M)
    var = var + W_S.pow(2).sum(dim=-1)
    return loc, var


# --------------------------------------------------------------------------
# parameter containers
# --------------------------------------------------------------------------
def grid_inducing_points(world: Sequence[Tuple[float, float]], n_points: Sequence[int], dtype=torch.float32):
    """sparse_gdrf.py:61-77 'grid' init + scale to the unit cube."""
    pts = [torch.arange(b[0], b[1] + (b[1] - b[0]) / (n - 1) - 1e-10, (b[1] - b[0]) / (n - 1))
           for b, n in zip(world, n_points)]
    Z = torch.stack([x.flatten() for x in torch.meshgrid(*pts, indexing="ij")]).T
    lo = torch.tensor([b[0] for b in world])
    hi = torch.tensor([b[1] for b in world])
    return ((Z - lo) / (hi - lo)).to(dtype)


def validate_dirichlet_param(b, K: int, V: int) -> torch.Tensor:
    b = torch.as_tensor(b, dtype=torch.float32)
    assert (b <= 0).sum().item() == 0, "b must be positive"
    if b.dim() == 0:
        return torch.ones(K, V) * b
    if b.dim() == 1:
        if b.shape[0] == K:
            return b.repeat(V, 1).T
        if b.shape[0] == V:
            return b.repeat(K, 1)
        raise ValueError("parameter b must have length K or V if 1D")
    if b.dim() == 2:
        assert b.shape == (K, V), "b should be KxV if 2D"
        return b
    raise ValueError("invalid b parameter")


class RefShapedGDRF:
    """Reference-shaped SparseMultinomialGDRF + SVI(Trace_ELBO) + per-parameter optimizers."""

    def __init__(self, xs, ws, *, kind="rbf", K=3, n_points=(8, 4), lengthscale=0.1, variance=25.0,
                 dirichlet_param=0.01, jitter=1e-8, maxjitter=15, noise=1.0, dtype=torch.float64,
                 Z: Optional[torch.Tensor] = None, optimizer="adam", lr=1e-3,
                 force_jitter_level: Optional[int] = None, learn_inducing: bool = False, scale_mixture: float = 1.0,
                 whiten: bool = True, mean_function=None, world: Optional[Sequence[Tuple[float, float]]] = None,
                 guide_rescale: bool = True, link_function=None):
        self.dtype = dtype
        # topic_model.py:148-198: inputs are mapped to the unit cube by (x - lower) / delta.  With world=None the inputs are taken as
        # already scaled (what train() builds: world = [(0, 1)]^D, train_script.py:261-271).  Quirk Q3 (sparse_gdrf.py:376-380): the
        # guide is decorated with @scale_decorator AND calls self.scale(xs) again, so for a non-unit world the guide's conditional
        # is evaluated at the DOUBLY scaled inputs while the model's sees the singly scaled ones; guide_rescale=True restates that.
        self.world = None if world is None else [(float(a), float(b)) for a, b in world]
        self.guide_rescale = bool(guide_rescale)
        self.mean_function = mean_function          # abstract_gdrf.py:33-48; None = zero_mean (abstract_gdrf.py:17-18)
        # abstract_gdrf.py:34-50: None = softmax_link_function (abstract_gdrf.py:21-22: softmax over dim -2)
        self.link_function = (lambda x: torch.softmax(x, -2)) if link_function is None else link_function
        self.kind = kind
        self.K = K
        self.xs = torch.as_tensor(xs).to(dtype)
        self.ws = torch.as_tensor(ws).to(torch.int32)
        self.N, self.D = self.xs.shape
        self.V = self.ws.shape[1]
        self.jitter, self.maxjitter = jitter, maxjitter
        self.force_jitter_level = force_jitter_level
        unit = [(0.0, 1.0)] * self.D
        if self.world is not None:
            self._lower = torch.tensor([b[0] for b in self.world], dtype=dtype)
            self._delta = torch.tensor([b[1] - b[0] for b in self.world], dtype=dtype)
        # sparse_gdrf.py:61-77: grid points are laid out in the world and then scaled: the same unit-cube grid
        self.Z = (grid_inducing_points(unit, list(n_points)) if Z is None else torch.as_tensor(Z)).to(dtype)
        self.M = self.Z.shape[0]
        self.alpha = validate_dirichlet_param(dirichlet_param, K, self.V).to(dtype)
        # unconstrained parameters (PyroParam storage)
        ls = torch.tensor(float(lengthscale), dtype=dtype)
        var = torch.tensor(float(variance), dtype=dtype)
        sm = torch.tensor(float(scale_mixture), dtype=dtype)
        Kuu = kernel_matrix(kind, self.Z, self.Z, ls, var, sm)
        L0, self.init_jitter_level = jittercholesky(Kuu, self.M, jitter, maxjitter,
                                                    force_level=force_jitter_level)
        wt = torch.softmax(self.alpha, dim=-2)           # abstract_gdrf.py:68-69 (over K)
        self.params: Dict[str, torch.Tensor] = {
            "log_lengthscale": ls.log().clone(),
            "log_variance": var.log().clone(),
            "u_loc": torch.zeros(K, self.M, dtype=dtype),
            "u_scale_tril_unc": transform_to(constraints.lower_cholesky).inv(L0.repeat(K, 1, 1)).clone(),
            "log_noise": torch.tensor(float(noise), dtype=dtype).log().clone(),
            "phi_unc": wt.log().clone(),                  # simplex transform inverse = log
        }
        if kind == "rationalquadratic":
            self.params["log_scale_mixture"] = sm.log().clone()          # positive constraint -> exp
        self.whiten = bool(whiten)
        self.learn_inducing = bool(learn_inducing)
        if self.learn_inducing:
            # sparse_gdrf.py:79-88: PyroParam(scaled points, constraint=stack([interval(0, 1)] * D)); the stored value is
            # transform_to(interval).inv(Z) = logit of Z clamped to [tiny, 1 - eps] (torch SigmoidTransform._inverse)
            self.params["inducing_unc"] = transform_to(constraints.interval(0.0, 1.0)).inv(self.Z).clone()
        for p in self.params.values():
            p.requires_grad_(True)
        self.optimizer_name = optimizer
        self.lr = lr
        self._opts: Dict[str, object] = {}
        self.last_jitter_level = None
        self.last_terms: Dict[str, float] = {}

    # ---- constrained views -------------------------------------------------
    def inducing(self) -> torch.Tensor:
        """The (M, D) inducing inputs in the scaled world: fixed, or the interval(0,1)-constrained parameter."""
        if self.learn_inducing:
            return transform_to(constraints.interval(0.0, 1.0))(self.params["inducing_unc"])
        return self.Z

    def constrained(self):
        p = self.params
        return dict(
            lengthscale=p["log_lengthscale"].exp(),
            variance=p["log_variance"].exp(),
            u_loc=p["u_loc"],
            u_scale_tril=transform_to(constraints.lower_cholesky)(p["u_scale_tril_unc"]),
            noise=p["log_noise"].exp(),
            phi=torch.softmax(p["phi_unc"], dim=-1),
            scale_mixture=p["log_scale_mixture"].exp() if "log_scale_mixture" in p else None,
        )

    def scale(self, xs: torch.Tensor) -> torch.Tensor:
        """topic_model.py:168-169 (identity when no world was given)."""
        return xs if self.world is None else (xs - self._lower) / self._delta

    def _luu(self, c):
        Zc = self.inducing()
        Kuu = kernel_matrix(self.kind, Zc, Zc, c["lengthscale"], c["variance"], c["scale_mixture"]).contiguous()
        L, lvl = jittercholesky(Kuu, self.M, self.jitter, self.maxjitter, force_level=self.force_jitter_level)
        self.last_jitter_level = lvl
        return L

    # ---- one ELBO evaluation, reference-shaped ------------------
    def loss(self, eps: torch.Tensor, xs=None, ws=None, n_global: Optional[int] = None) -> torch.Tensor:
        xs = self.xs if xs is None else torch.as_tensor(xs).to(self.dtype)
        ws = self.ws if ws is None else torch.as_tensor(ws).to(torch.int32)
        scale = 1.0 / (n_global if n_global is not None else self.N)   # train_script.py:365 (Q9)
        eps = torch.as_tensor(eps).to(self.dtype)
        c = self.constrained()
        xs = self.scale(xs)                                # @scale_decorator("xs") on model and guide
        xs_g = self.scale(xs) if self.guide_rescale else xs   # the guide's own `xs = self.scale(xs)` (quirk Q3)
        # guide: sparse_gdrf.py:375-409
        Luu = self._luu(c)
        f_loc, f_var = conditional(self.kind, xs_g, self.inducing(), c["lengthscale"], c["variance"],
                                   c["u_loc"], c["u_scale_tril"], Luu, c["scale_mixture"], self.whiten)
        if self.mean_function is not None:
            f_loc = f_loc + self.mean_function(xs_g)      # sparse_gdrf.py:395
        q_mu = Normal(f_loc, f_var)                       # Q1: variance passed as scale
        mu = f_loc + f_var * eps                          # rsample with injected eps
        lq_mu = q_mu.log_prob(mu).sum()
        # model (replayed with mu, phi): sparse_gdrf.py:323-373
        Luu2 = self._luu(c)
        f_loc2, f_var2 = conditional(self.kind, xs, self.inducing(), c["lengthscale"], c["variance"],
                                     c["u_loc"], c["u_scale_tril"], Luu2, c["scale_mixture"], self.whiten)
        if self.mean_function is not None:
            f_loc2 = f_loc2 + self.mean_function(xs)      # sparse_gdrf.py:346
        lp_mu = Normal(f_loc2, f_var2 + c["noise"]).log_prob(mu).sum()
        lp_phi = Dirichlet(self.alpha).log_prob(c["phi"]).sum()
        topic_probs = self.link_function(mu).transpose(-2, -1)          # sparse_gdrf.py:361
        probs = torch.matmul(topic_probs, c["phi"])
        ll = Multinomial(probs=probs, validate_args=False).log_prob(ws).sum()
        elbo = scale * (lp_mu + lp_phi + ll - lq_mu)
        self.last_terms = dict(lp_mu=float(lp_mu.detach()), lq_mu=float(lq_mu.detach()), lp_phi=float(lp_phi.detach()),
                               ll=float(ll.detach()))
        return -elbo

    def row_terms(self, eps: torch.Tensor, xs, ws, Luu=None) -> torch.Tensor:
        """Sum over the given rows of every per-observation term of the ELBO, unscaled: lp_mu + ll - lq_mu.
        The guide's and the model's ``conditional`` have identical values (quirk Q4), so it is evaluated once here; used by
        :meth:`loss_chunked` to evaluate sizes whose N x M x K intermediate does not fit the host at once."""
        xs = torch.as_tensor(xs).to(self.dtype)
        ws = torch.as_tensor(ws).to(torch.int32)
        eps = torch.as_tensor(eps).to(self.dtype)
        c = self.constrained()
        Luu = self._luu(c) if Luu is None else Luu
        f_loc, f_var = conditional(self.kind, xs, self.inducing(), c["lengthscale"], c["variance"],
                                   c["u_loc"], c["u_scale_tril"], Luu, c["scale_mixture"], self.whiten)
        if self.mean_function is not None:
            f_loc = f_loc + self.mean_function(xs)
        mu = f_loc + f_var * eps
        lq_mu = Normal(f_loc, f_var).log_prob(mu).sum()
        lp_mu = Normal(f_loc, f_var + c["noise"]).log_prob(mu).sum()
        probs = torch.matmul(self.link_function(mu).transpose(-2, -1), c["phi"])
        ll = Multinomial(probs=probs, validate_args=False).log_prob(ws).sum()
        return lp_mu + ll - lq_mu

    def loss_chunked(self, eps, xs, ws, n_global: int, chunk: int = 25000, grad_names=("u_loc", "phi_unc")):
        """The loss of :meth:`loss` evaluated in row chunks (every term but the Dirichlet one is a sum over observations),
        with autograd gradients of the parameters in ``grad_names`` accumulated chunk by chunk (only parameters the
        predictive variance does not depend on are cheap: u_loc, phi_unc, log_noise).  Returns (loss, {name: grad})."""
        N = xs.shape[0]
        c = self.constrained()
        with torch.no_grad():
            Luu = self._luu(c)
        total = 0.0
        grads = {k: torch.zeros_like(self.params[k]) for k in grad_names}
        ps = [self.params[k] for k in grad_names]
        for lo in range(0, N, chunk):
            hi = min(N, lo + chunk)
            t = self.row_terms(eps[:, lo:hi], xs[lo:hi], ws[lo:hi], Luu=Luu)
            for k, g in zip(grad_names, torch.autograd.grad(t, ps, allow_unused=True)):
                if g is not None:
                    grads[k] += g
            total += float(t.detach())
        lp_phi = Dirichlet(self.alpha).log_prob(self.constrained()["phi"]).sum()
        if "phi_unc" in grads:
            grads["phi_unc"] += torch.autograd.grad(lp_phi, self.params["phi_unc"])[0]
        scale = 1.0 / n_global
        return -scale * (total + float(lp_phi.detach())), {k: -scale * g for k, g in grads.items()}

    def renyi_loss(self, eps: torch.Tensor, alpha: float = 0.0, **kw) -> torch.Tensor:
        """pyro.infer.RenyiELBO(alpha, num_particles=P) (pyro-ppl 1.8.0, third-party; published estimator, Li & Turner 2016):
        with the scaled per-particle ELBOs e_p,  loss = -(logsumexp((1 - alpha) e_p) - log P) / (1 - alpha).  All sites are
        reparameterised, so pyro's surrogate gradient (normalised detached weights times grad e_p) is the gradient of this."""
        eps = torch.as_tensor(eps)
        assert eps.dim() == 3 and alpha != 1.0
        e = torch.stack([-self.loss(eps[p], **kw) for p in range(eps.shape[0])])
        return -(torch.logsumexp((1.0 - alpha) * e, 0) - math.log(eps.shape[0])) / (1.0 - alpha)

    def loss_and_grads(self, eps, renyi_alpha: Optional[float] = None, **kw) -> Tuple[float, Dict[str, torch.Tensor]]:
        for p in self.params.values():
            p.grad = None
        loss = self.loss(eps, **kw) if renyi_alpha is None else self.renyi_loss(eps, renyi_alpha, **kw)
        loss.backward()
        return float(loss.detach()), {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                             for k, p in self.params.items()}

    # ---- optimizers: one instance per parameter -----------------
    def _get_opt(self, name, p):
        if name in self._opts:
            return self._opts[name]
        if self.optimizer_name == "adam":
            o = torch.optim.Adam([p], lr=self.lr)
        elif self.optimizer_name == "adamw":
            o = torch.optim.AdamW([p], lr=self.lr)
        elif self.optimizer_name == "clippedadam":
            o = _ClippedAdam(p, lr=self.lr)
        else:
            raise ValueError(self.optimizer_name)
        self._opts[name] = o
        return o

    def step(self, eps, **kw) -> float:
        """SVI.step: loss_and_grads, per-parameter optimizer step, zero grads."""
        loss, _ = self.loss_and_grads(eps, **kw)
        with torch.no_grad():
            pass
        for name, p in self.params.items():
            self._get_opt(name, p).step()
        for p in self.params.values():
            p.grad = None
        return loss

    # ---- predictive path (sparse_gdrf.py:161-186, abstract_gdrf.py:113-139) --
    @torch.no_grad()
    def log_topic_probs(self, xs=None):
        xs = self.scale(self.xs if xs is None else torch.as_tensor(xs).to(self.dtype))     # scaled once (sparse_gdrf.py:161-162)
        c = self.constrained()
        Luu = self._luu(c)
        f_loc, _ = conditional(self.kind, xs, self.inducing(), c["lengthscale"], c["variance"],
                               c["u_loc"], c["u_scale_tril"], Luu, c["scale_mixture"], self.whiten)
        return f_loc

    def topic_probs(self, xs=None):
        return self.link_function(self.log_topic_probs(xs)).T            # abstract_gdrf.py:113-115

    def word_probs(self, xs=None):
        return self.topic_probs(xs) @ self.constrained()["phi"].detach()

    def perplexity(self, xs=None, ws=None):
        ws = self.ws if ws is None else torch.as_tensor(ws)
        w = ws.to(self.dtype)
        return ((w * self.word_probs(xs).log()).sum() / -w.sum()).exp()


class _ClippedAdam:
    """pyro.optim.ClippedAdam: clip +-10, lrd 1.0, eps outside the bias correction."""

    def __init__(self, p, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clip_norm=10.0, lrd=1.0):
        self.p, self.lr, self.betas, self.eps, self.clip, self.lrd = p, lr, betas, eps, clip_norm, lrd
        self.m = torch.zeros_like(p)
        self.v = torch.zeros_like(p)
        self.t = 0

    @torch.no_grad()
    def step(self):
        g = self.p.grad.clamp(-self.clip, self.clip)
        self.t += 1
        self.lr *= self.lrd
        b1, b2 = self.betas
        self.m.mul_(b1).add_(g, alpha=1 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
        step_size = self.lr * math.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
        self.p.addcdiv_(self.m, self.v.sqrt().add_(self.eps), value=-step_size)


# --------------------------------------------------------------------------
# fused-shaped evaluation with the hand-derived backward,
# in the factorisation the HIP kernels use.  numpy, any float dtype.
# --------------------------------------------------------------------------
def _np_kernel(kind, X, Z, ls, var):
    """Direct (x-z)^2 form (what the HIP kernels compute).  Returns (k, r2)."""
    d = X[:, None, :] - Z[None, :, :]
    r2 = (d * d).sum(-1) / (ls * ls)
    if kind == "rbf":
        return var * np.exp(-0.5 * r2), r2
    r = np.sqrt(r2 + 1e-12)
    if kind == "exponential":
        return var * np.exp(-r), r2
    if kind == "matern32":
        a = 3.0 ** 0.5 * r
        return var * (1 + a) * np.exp(-a), r2
    a = SQRT5 * r
    return var * (1 + a + (5.0 / 3.0) * r * r) * np.exp(-a), r2


def _np_dk_dlogls(kind, k, r2, var):
    if kind == "rbf":
        return k * r2
    r = np.sqrt(r2 + 1e-12)
    if kind == "exponential":
        return k * (r2 / r)
    if kind == "matern32":
        a = 3.0 ** 0.5 * r
        return var * np.exp(-a) * a * 3.0 ** 0.5 * (r2 / r)
    a = SQRT5 * r
    return var * np.exp(-a) * (a / 3.0) * (1 + a) * SQRT5 * (r2 / r)


def _np_transforms(params):
    K = params["u_loc"].shape[0]
    M = params["u_loc"].shape[1]
    Sunc = params["u_scale_tril_unc"]
    S = np.tril(Sunc, -1) + np.stack([np.diag(np.exp(np.diag(Sunc[k]))) for k in range(K)])
    phi_unc = params["phi_unc"]
    phi = np.exp(phi_unc - phi_unc.max(-1, keepdims=True))
    phi = phi / phi.sum(-1, keepdims=True)
    return (np.exp(params["log_lengthscale"]), np.exp(params["log_variance"]), np.exp(params["log_noise"]), S, phi, K, M)


def fused_local(kind, xs, ws, Z, params: Dict[str, np.ndarray], eps, jitter_total_: float, heavy: bool = True):
    """Everything of one step that is a sum over THIS shard's observations: the all-reduce payload
    (what gdrf_step_local writes to red_T / red_d) plus the stage values (aux).  heavy=False stops after the row-local
    backward (no Wbar / A_k / G / K_nm sums: the N M^2 K backward contractions), which is what a chunked evaluation of the
    headline size can afford on the host; the payload then lacks those entries."""
    xs = np.asarray(xs)
    dt = xs.dtype
    ws_f = np.asarray(ws).astype(dt)
    ls, var, eta, S, phi, K, M = _np_transforms(params)
    U = params["u_loc"]
    Kuu0, _ = _np_kernel(kind, Z, Z, ls, var)
    Kuu = Kuu0 + jitter_total_ * np.eye(M, dtype=dt)
    L = np.linalg.cholesky(Kuu)
    Linv = np.linalg.inv(L)
    Knm, R2nm = _np_kernel(kind, xs, Z, ls, var)
    W = Knm @ Linv.T
    q = (W * W).sum(-1)
    loc = U @ W.T                                         # (K,N)
    T = np.matmul(W[None], S)                             # (K,N,M): T_k = W S_k
    tt = (T * T).sum(-1)
    a = (var - q > 0).astype(dt)
    v = a * (var - q) + tt
    s = v + eta
    r = v / s
    mu = loc + v * eps
    mx = mu.max(0, keepdims=True)
    e = np.exp(mu - mx)
    theta = e / e.sum(0, keepdims=True)                   # (K,N)
    p = theta.T @ phi                                     # (N,V)
    ps = p.sum(-1, keepdims=True)
    phat = p / ps
    feps = np.finfo(dt).eps
    mask = ((phat > feps) & (phat < 1 - feps)).astype(dt)
    logit = np.log(np.clip(phat, feps, 1 - feps))
    from scipy.special import gammaln
    ll_const = float((gammaln(ws_f.sum(-1).astype(np.float64) + 1) - gammaln(ws_f.astype(np.float64) + 1).sum(-1)).sum())
    llw = float((ws_f * logit).sum())
    c_site = -np.log(s) + np.log(v) - 0.5 * eps ** 2 * r ** 2 + 0.5 * eps ** 2
    pbar = ws_f * mask / p
    thbar = phi @ pbar.T                                  # (K,N)
    mubar = theta * (thbar - (theta * thbar).sum(0, keepdims=True))
    dc_dv = -1 / s + 1 / v - eps ** 2 * r * eta / s ** 2
    dc_deta = -1 / s + eps ** 2 * r ** 2 / s
    vbar = mubar * eps + dc_dv
    locbar = mubar
    if not heavy:
        payload = dict(site=float(c_site.sum()), llw=llw, ll_const=ll_const, noise_g=float(dc_deta.sum()),
                       var_direct=float((a * vbar.sum(0)).sum()), ubar=locbar @ W, phibar_lik=theta @ pbar)
        return payload, dict(q=q, loc=loc, tt=tt, var=v, mu=mu, theta=theta, vbar=vbar, locbar=locbar, L=L, Linv=Linv, phi=phi, S=S)
    B = np.einsum("kij,klj->kil", S, S)                   # S_k S_k^T
    Wbar = locbar.T @ U - 2 * (a * vbar.sum(0))[:, None] * W
    for k in range(K):
        Wbar += (2 * vbar[k])[:, None] * (W @ B[k])
    Knm_bar = Wbar @ Linv
    payload = dict(
        site=float(c_site.sum()), llw=llw, ll_const=ll_const, noise_g=float(dc_deta.sum()),
        var_direct=float((a * vbar.sum(0)).sum()), knm_k=float((Knm_bar * Knm).sum()),
        knm_dls=float((Knm_bar * _np_dk_dlogls(kind, Knm, R2nm, var)).sum()),
        ubar=locbar @ W, phibar_lik=theta @ pbar, A=np.einsum("ni,kn,nj->kij", W, vbar, W), G=Wbar.T @ W)
    aux = dict(W=W, q=q, loc=loc, tt=tt, var=v, mu=mu, theta=theta, vbar=vbar, locbar=locbar, Wbar=Wbar, Knm=Knm, Kuu=Kuu,
               L=L, Linv=Linv, phi=phi, S=S, B=B)
    return payload, aux


def fused_finish(kind, Z, params: Dict[str, np.ndarray], alpha, payload, n_global: float, jitter_total_: float):
    """Replicated epilogue on the (all-reduced) payload: what gdrf_step_finish computes."""
    dt = Z.dtype
    ls, var, eta, S, phi, K, M = _np_transforms(params)
    from scipy.special import gammaln
    Kuu0, R2uu = _np_kernel(kind, Z, Z, ls, var)
    L = np.linalg.cholesky(Kuu0 + jitter_total_ * np.eye(M, dtype=dt))
    Linv = np.linalg.inv(L)
    a64 = np.asarray(alpha, dtype=np.float64)
    lp_phi = float((gammaln(a64.sum(-1)) - gammaln(a64).sum(-1) + ((a64 - 1) * np.log(phi)).sum(-1)).sum())
    elbo_n = payload["site"] + payload["llw"] + payload["ll_const"] + lp_phi
    loss = -elbo_n / n_global
    Sbar = 2 * np.einsum("kij,kjl->kil", payload["A"], S)
    Lbar = -np.tril(Linv.T @ payload["G"])
    Pm = np.tril(L.T @ Lbar)
    Pm[np.diag_indices(M)] *= 0.5
    Sp = Linv.T @ Pm @ Linv
    Kuu_bar = 0.5 * (Sp + Sp.T)
    g_logvar = payload["knm_k"] + float((Kuu_bar * Kuu0).sum()) + var * payload["var_direct"]
    g_logls = payload["knm_dls"] + float((Kuu_bar * _np_dk_dlogls(kind, Kuu0, R2uu, var)).sum())
    g_lognoise = float(eta * payload["noise_g"])
    g_Sunc = np.tril(Sbar, -1)
    for k in range(K):
        g_Sunc[k][np.diag_indices(M)] = np.diag(Sbar[k]) * np.diag(S[k])
    phibar = payload["phibar_lik"] + (np.asarray(alpha, dtype=dt) - 1) / phi
    g_phiunc = phi * (phibar - (phi * phibar).sum(-1, keepdims=True))
    sc = -1.0 / n_global
    grads = dict(
        log_lengthscale=np.asarray(sc * g_logls), log_variance=np.asarray(sc * g_logvar),
        u_loc=sc * payload["ubar"], u_scale_tril_unc=sc * g_Sunc, log_noise=np.asarray(sc * g_lognoise),
        phi_unc=sc * g_phiunc,
    )
    return loss, grads, dict(lp_phi=lp_phi, elbo_n=elbo_n, Kuu_bar=Kuu_bar)


def fused_elbo_and_grads(kind, xs, ws, Z, params: Dict[str, np.ndarray], alpha, eps, jitter_total_: float,
                         n_global: Optional[int] = None):
    """Returns (loss, grads-of-loss wrt unconstrained params, aux) in the factorisation of the HIP kernels:
       W = Knm Linv^T ; T_k = W S_k (forward variance) ;
       Wbar = sum_k diag(2 vbar_k) W B_k (+ loc and clamp terms), B_k = S_k S_k^T ;
       A_k = W^T diag(vbar_k) W ; Sbar_k = 2 A_k S_k ; G = Wbar^T W ; Lbar = -tril(Linv^T G) ;
       Cholesky backward through the explicit inverse."""
    payload, aux = fused_local(kind, xs, ws, Z, params, eps, jitter_total_)
    ng = float(n_global if n_global is not None else np.asarray(xs).shape[0])
    loss, grads, fin = fused_finish(kind, np.asarray(Z), params, alpha, payload, ng, jitter_total_)
    aux.update(A=payload["A"], G=payload["G"], ll=payload["llw"] + payload["ll_const"], ll_const=payload["ll_const"],
               site=payload["site"], **fin)
    return loss, grads, aux


# synthetic data lives with the product's drivers (bench / examples); re-exported for the tests
from gdrf_amd.data import synth_circles  # noqa: E402,F401

"""CPU tests of the oracle: analytic known-answer cases, the hand-derived backward
against autograd, and the committed golden fixtures (regression pin of the oracle itself)."""
import os

import numpy as np
import pytest
import torch

from oracle.gdrf_oracle import (RefShapedGDRF, conditional, fused_elbo_and_grads, grid_inducing_points, jitter_total,
                                jittercholesky, kernel_matrix, validate_dirichlet_param)
from gdrf_amd.data import synth_circles
from tests._util import make_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fresh(K=3, V=12, **kw):
    xs, ws, _ = synth_circles(10, 6, V, K, seed=4)
    return RefShapedGDRF(xs, ws, K=K, n_points=(4, 3), dtype=torch.float64, jitter=1e-6, lengthscale=0.3, **kw)


def test_kat1_zero_u_loc_gives_uniform_topic_probs():
    m = _fresh()
    assert torch.allclose(m.log_topic_probs(), torch.zeros(m.K, m.N, dtype=torch.float64), atol=0)
    assert torch.allclose(m.topic_probs(), torch.full((m.N, m.K), 1.0 / m.K, dtype=torch.float64), atol=1e-15)


def test_kat2_initial_word_topic_matrix_is_uniform_and_perplexity_is_V():
    m = _fresh(V=12)
    phi = m.constrained()["phi"]
    assert torch.allclose(phi, torch.full_like(phi, 1.0 / m.V), atol=1e-15)       # quirk Q2
    assert abs(float(m.perplexity()) - m.V) < 1e-10
    # the log-likelihood is then independent of eps and of the GP parameters
    g = torch.Generator().manual_seed(0)
    m.loss(torch.randn(m.K, m.N, generator=g, dtype=torch.float64))
    ll1 = m.last_terms["ll"]
    m.loss(torch.randn(m.K, m.N, generator=g, dtype=torch.float64))
    assert abs(ll1 - m.last_terms["ll"]) < 1e-8 * abs(ll1)
    w = m.ws.double()
    expect = (torch.lgamma(w.sum(-1) + 1) - torch.lgamma(w + 1).sum(-1)).sum() - np.log(m.V) * w.sum()
    assert abs(ll1 - float(expect)) < 1e-6 * abs(float(expect))


@pytest.mark.parametrize("kind", ["rbf", "matern52", "matern32", "exponential"])
def test_kat3_kernel_at_coincident_points_is_the_variance(kind):
    Z = grid_inducing_points([(0, 1), (0, 1)], [3, 3], dtype=torch.float64)
    K = kernel_matrix(kind, Z, Z, torch.tensor(0.2, dtype=torch.float64), torch.tensor(7.0, dtype=torch.float64))
    assert torch.allclose(K.diagonal(), torch.full((9,), 7.0, dtype=torch.float64), atol=1e-4)
    assert torch.allclose(K, K.T, atol=1e-12)


def test_kat4_initial_variance_formula():
    m = _fresh()
    c = m.constrained()
    L = m._luu(c)
    loc, var = conditional(m.kind, m.xs, m.Z, c["lengthscale"], c["variance"], c["u_loc"], c["u_scale_tril"], L)
    Kzx = kernel_matrix(m.kind, m.Z, m.xs, c["lengthscale"], c["variance"])
    W = torch.linalg.solve_triangular(L, Kzx, upper=False).T
    expect = (c["variance"] - (W ** 2).sum(-1)).clamp(min=0) + ((W @ L) ** 2).sum(-1)     # S_k = L_uu at init
    assert torch.allclose(var, expect.expand(m.K, -1), rtol=1e-10, atol=1e-10)
    assert float(loc.detach().abs().max()) == 0.0


def test_kat5_predict_at_inducing_points():
    Z = grid_inducing_points([(0, 1), (0, 1)], [3, 2], dtype=torch.float64)
    ls, var = torch.tensor(0.5, dtype=torch.float64), torch.tensor(3.0, dtype=torch.float64)
    jit = 1e-3
    Kuu = kernel_matrix("rbf", Z, Z, ls, var) + jit * torch.eye(6, dtype=torch.float64)
    L = torch.linalg.cholesky(Kuu)
    u = torch.randn(2, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    S = torch.eye(6, dtype=torch.float64).repeat(2, 1, 1)
    loc, v = conditional("rbf", Z, Z, ls, var, u, S, L)
    # W = (L^-1 (Kuu - jit I))^T : without jitter W = L exactly; with it ||w||^2 = Kuu0 L^-T L^-1 Kuu0 diag
    W = torch.linalg.solve_triangular(L, Kuu - jit * torch.eye(6, dtype=torch.float64), upper=False).T
    assert torch.allclose(loc, u @ W.T, atol=1e-12)
    assert torch.allclose(v, (var - (W ** 2).sum(-1)).clamp(min=0) + (W ** 2).sum(-1), atol=1e-12)


def test_kat6_site_terms_cancel_when_noise_vanishes():
    m = _fresh()
    with torch.no_grad():
        m.params["log_noise"].fill_(-80.0)
    g = torch.Generator().manual_seed(0)
    m.loss(torch.randn(m.K, m.N, generator=g, dtype=torch.float64))
    assert abs(m.last_terms["lp_mu"] - m.last_terms["lq_mu"]) < 1e-6 * abs(m.last_terms["lq_mu"])


def test_kat8_cumulative_jitter_schedule_on_singular_kuu():
    Z = grid_inducing_points([(0, 1), (0, 1)], [3, 3], dtype=torch.float32)
    Z[1] = Z[0]
    Kuu = kernel_matrix("rbf", Z, Z, torch.tensor(0.3), torch.tensor(25.0))
    L, lvl = jittercholesky(Kuu, 9, 1e-8, 15)
    assert lvl >= 1 and torch.isfinite(L).all()
    assert abs(jitter_total(1e-8, lvl) - 1e-8 * (10 ** (lvl + 1) - 1) / 9) < 1e-20 + 1e-12 * jitter_total(1e-8, lvl)
    with pytest.raises(RuntimeError, match="reached max jitter"):
        jittercholesky(kernel_matrix("rbf", Z.double(), Z.double(), torch.tensor(0.3, dtype=torch.float64),
                                     torch.tensor(25.0, dtype=torch.float64)), 9, 1e-30, 3)


def test_validate_dirichlet_param_shapes():
    assert validate_dirichlet_param(0.5, 3, 4).shape == (3, 4)
    assert torch.equal(validate_dirichlet_param(torch.tensor([1., 2., 3.]), 3, 4)[:, 0], torch.tensor([1., 2., 3.]))
    assert torch.equal(validate_dirichlet_param(torch.tensor([1., 2., 3., 4.]), 3, 4)[0], torch.tensor([1., 2., 3., 4.]))
    with pytest.raises(ValueError):
        validate_dirichlet_param(torch.ones(5), 3, 4)
    with pytest.raises(AssertionError):
        validate_dirichlet_param(-1.0, 3, 4)


@pytest.mark.parametrize("kind", ["rbf", "matern52", "matern32", "exponential"])
def test_hand_derived_backward_matches_autograd(kind):
    m, eps = make_oracle(kind=kind, W=12, H=5, V=20, K=4, n_points=(4, 3))
    loss, grads = m.loss_and_grads(eps)
    P = {k: v.detach().numpy().copy() for k, v in m.params.items()}
    l2, g2, _ = fused_elbo_and_grads(kind, m.xs.numpy(), m.ws.numpy(), m.Z.numpy(), P, m.alpha.numpy(), eps.numpy(),
                                     jitter_total(m.jitter, m.last_jitter_level))
    assert abs(loss - l2) < 1e-6 * abs(loss)         # torch evaluates lgamma(int32 counts) in float32
    for k in grads:
        a, b = grads[k].numpy(), g2[k]
        # the oracle keeps pyro's expanded-form distance, the fused version the direct (x-z)^2 (quirk Q12): at coincident
        # points sqrt(r2 + 1e-12) sees the ~1e-16 cancellation noise, which the non-smooth kernels pass on at ~1e-8 relative
        tol = 1e-10 if kind in ("rbf", "matern52") else 1e-7
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), k


def test_scale_uses_global_n_for_minibatches():
    """Q9: a streaming mini-batch is scaled by 1/len(full data)."""
    m, eps = make_oracle(W=12, H=5)
    full = float(m.loss(eps).detach())
    sub = float(m.loss(eps[:, :10], xs=m.xs[:10], ws=m.ws[:10], n_global=m.N).detach())
    sub_default = float(m.loss(eps[:, :10], xs=m.xs[:10], ws=m.ws[:10]).detach())      # default scale: 1/len(full data)
    sub_local = float(m.loss(eps[:, :10], xs=m.xs[:10], ws=m.ws[:10], n_global=10).detach())
    assert sub_default == sub
    assert abs(sub_local / sub - m.N / 10) < 1e-9 * m.N
    assert np.isfinite(full)


@pytest.mark.parametrize("name", ["g1_artificial2d_rbf.npz", "g2_synth1d_matern52.npz"])
def test_oracle_reproduces_golden_runs(name):
    g = np.load(os.path.join(GOLD, name))
    kind = "rbf" if "rbf" in name else "matern52"
    K = g["p0_u_loc"].shape[0]
    m = RefShapedGDRF(g["xs"], g["ws"], kind=kind, K=K, Z=torch.from_numpy(g["Z"]), dtype=torch.float64,
                      jitter=float(g["jitter"]), optimizer="adam", lr=1e-2)
    for k in m.params:
        with torch.no_grad():
            m.params[k].copy_(torch.from_numpy(g["p0_" + k]))
    m.alpha = torch.from_numpy(g["alpha"])
    eps = torch.from_numpy(g["eps"])
    loss, grads = m.loss_and_grads(eps[0])
    assert abs(loss - float(g["loss0"])) < 1e-12 * abs(loss)
    for k in grads:
        assert np.abs(grads[k].numpy() - g["g0_" + k]).max() < 1e-12
    losses = [m.step(eps[s]) for s in range(eps.shape[0])]
    assert np.allclose(losses, g["losses"], rtol=1e-11)
    assert np.allclose(m.topic_probs().numpy(), g["topic_probs"], atol=1e-11)


def test_learnable_inducing_inputs_oracle():
    """fixed_inducing_points=False (sparse_gdrf.py:79-88): Z = interval(0,1)-transform of an unconstrained block.  The
    constrained value starts at the given points, and the autograd gradient w.r.t. the block agrees with a central
    finite difference of the loss (pins the oracle path the GPU test compares against)."""
    from tests._util import make_oracle
    m, eps = make_oracle(dtype=torch.float64, learn_inducing=True, random_inducing=True, W=10, H=8, V=6, K=2, n_points=(3, 3),
                         jitter=1e-6, lengthscale=0.3)
    assert torch.allclose(m.inducing().detach(), m.Z, atol=1e-14)
    _, g = m.loss_and_grads(eps)
    for idx in [(0, 0), (4, 1), (8, 0)]:
        h = 1e-6
        with torch.no_grad():
            m.params["inducing_unc"][idx] += h
        lp = float(m.loss(eps))
        with torch.no_grad():
            m.params["inducing_unc"][idx] -= 2 * h
        lm = float(m.loss(eps))
        with torch.no_grad():
            m.params["inducing_unc"][idx] += h
        fd = (lp - lm) / (2 * h)
        assert abs(fd - float(g["inducing_unc"][idx])) < 1e-6 * max(1.0, abs(fd)), (idx, fd, float(g["inducing_unc"][idx]))

"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import numpy as np
import torch

from oracle.gdrf_oracle import RefShapedGDRF, fused_elbo_and_grads, jitter_total, synth_circles

NAME_MAP = dict(log_lengthscale="log_lengthscale", log_variance="log_variance", log_noise="log_noise",
                u_loc="u_loc", phi_unc="phi_unc", u_scale_tril_unc="u_scale_tril_unc")


def make_oracle(kind="rbf", W=16, H=9, V=20, K=4, n_points=(4, 3), dtype=torch.float64, seed=1, jitter=1e-6,
                perturb=True, one_d=False, optimizer="adam", lr=1e-3, force_jitter_level=None, lengthscale=0.1,
                learn_inducing=False, random_inducing=False, scale_mixture=1.0, whiten=True, mean_function=None, s_perturb=0.1, trained_scale=None):
    xs, ws, _ = synth_circles(W, H, V, K, seed=seed, one_d=one_d)
    g = torch.Generator().manual_seed(seed + 100)
    Z = None
    if random_inducing:          # interior points: grid points on the world's boundary have a vanishing sigmoid Jacobian
        M = int(np.prod(n_points)) if not one_d else int(n_points[0])
        Z = 0.05 + 0.9 * torch.rand(M, 1 if one_d else 2, generator=g, dtype=torch.float64)
    m = RefShapedGDRF(xs, ws, kind=kind, K=K, n_points=n_points, dtype=dtype, jitter=jitter, optimizer=optimizer, lr=lr,
                      force_jitter_level=force_jitter_level, lengthscale=lengthscale, Z=Z, learn_inducing=learn_inducing,
                      scale_mixture=scale_mixture, whiten=whiten, mean_function=mean_function)
    if perturb:
        with torch.no_grad():
            m.params["u_loc"].add_(0.3 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64).to(dtype))
            if trained_scale is not None:
                # a posterior that has contracted, as after training: S_k = trained_scale I + small lower-triangular noise (the
                # initial S_k = L_uu of sparse_gdrf.py:100-110 makes tt = |S^T w|^2 - hence mu = loc + v eps - as large as cond(K_uu))
                u = s_perturb * torch.randn(m.params["u_scale_tril_unc"].shape, generator=g, dtype=torch.float64).tril(-1)
                u = u + torch.diag_embed(torch.full(m.params["u_loc"].shape, float(np.log(trained_scale)), dtype=torch.float64))
                m.params["u_scale_tril_unc"].copy_(u.to(dtype))
            else:
                m.params["u_scale_tril_unc"].add_(
                    s_perturb * torch.randn(m.params["u_scale_tril_unc"].shape, generator=g, dtype=torch.float64).tril().to(dtype))
            m.params["phi_unc"].add_(0.5 * torch.randn(m.params["phi_unc"].shape, generator=g, dtype=torch.float64).to(dtype))
            m.params["log_noise"].add_(0.2)
    eps = torch.randn(K, m.N, generator=g, dtype=torch.float64).to(dtype)
    return m, eps


def engine_from_oracle(m, dtype=None, device="cuda:0", n_cap=None, pure_fp32=False, store_t="auto", mfma_mode="auto", hyper_backward="auto"):
    """gdrf_amd.Engine holding exactly the oracle's parameters / inducing points / Dirichlet prior."""
    from gdrf_amd.engine import Engine
    dtype = m.dtype if dtype is None else dtype
    learn = bool(getattr(m, "learn_inducing", False))
    eng = Engine(n_cap or m.N, m.M, m.K, m.V, m.D, dtype=dtype, kernel=m.kind, device=device, jitter=m.jitter,
                 maxjitter=m.maxjitter, process_group=None, pure_fp32=pure_fp32, store_t=store_t, mfma_mode=mfma_mode,
                 learn_inducing=learn, whiten=bool(getattr(m, "whiten", True)), hyper_backward=hyper_backward)
    eng.set_inducing_points(m.Z)
    eng.set_dirichlet(m.alpha)
    load_params(eng, m)
    return eng


def load_params(eng, m):
    for name in eng.PARAM_NAMES:
        eng.view(name).copy_(m.params[name].detach().to(eng.dtype))
    for name in ("inducing_unc", "log_scale_mixture"):          # blocks only some configurations carry
        if name in m.params:
            eng.view(name).copy_(m.params[name].detach().to(eng.dtype))


def dev(t, eng, dtype=None):
    return torch.as_tensor(t).to(device=eng.device, dtype=dtype or eng.dtype).contiguous()


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))

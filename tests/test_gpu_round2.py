"""GPU parity tests added in round 2: the streaming mini-batch regime (gdrf/train_script.py:394-465), optimizer details,
shapes beyond the first round's limits, and the packed all-reduce payload."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests._util import dev, engine_from_oracle, make_oracle, relerr
from gdrf_amd import _lib

pytestmark = pytest.mark.gpu
LOSS_TOL_VS_TORCH = 1e-6          # torch evaluates lgamma(int32 counts) in float32 (tests/test_gpu_parity.py)


@pytest.mark.parametrize("n", [1, 3, 64])
def test_streaming_minibatch_steps_follow_the_oracle(n):
    """svi.step on n rows with the 1/len(full data) scale (quirk Q9), five consecutive steps on DIFFERENT same-shaped
    mini-batches: every loss (including the data-only Multinomial constant of that batch) and the parameters after the
    five Adam steps match the reference-shaped oracle."""
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6, optimizer="adam", lr=1e-2, W=24, H=16, V=15, K=3, n_points=(5, 4))
    eng = engine_from_oracle(m, n_cap=64)
    g = torch.Generator().manual_seed(11)
    N = m.N
    for step in range(5):
        sel = torch.randperm(N, generator=g)[:n]
        xs_b, ws_b = m.xs[sel], m.ws[sel]
        eps = torch.randn(m.K, n, generator=g, dtype=torch.float64)
        loss_ref = m.step(eps, xs=xs_b, ws=ws_b, n_global=N)
        # fresh device tensors every step, as train_script.py:461-465 builds them: the allocator may hand back the same storage
        eng.loss_and_grads(dev(xs_b, eng), dev(ws_b, eng, torch.int32), dev(eps, eng), n_global=N)
        eng.adam("adam", 1e-2)
        out = eng.read_out()
        assert abs(out["loss"] - loss_ref) <= LOSS_TOL_VS_TORCH * abs(loss_ref), (n, step, out["loss"], loss_ref)
    for name in eng.PARAM_NAMES:
        assert relerr(eng.view(name).cpu().numpy(), m.params[name].detach().numpy()) < 1e-7, (n, name)


def test_two_same_shaped_minibatches_get_their_own_multinomial_constant():
    """the data-only constant was cached by (data_ptr, version, shape); a second mini-batch living in recycled
    storage took the first one's constant.  Through the model surface, evaluate_loss on two batches, each against the oracle."""
    from gdrf_amd import poutine
    from gdrf_amd.infer import SVI, Trace_ELBO
    from gdrf_amd.kernels import RBF
    from gdrf_amd.models import SparseMultinomialGDRF
    from gdrf_amd.optim import Adam
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6, perturb=False, W=20, H=10, V=12, K=3, n_points=(4, 3), lengthscale=0.2)
    device = "cuda:0"
    xs, ws = m.xs.to(device), m.ws.to(device)
    model = SparseMultinomialGDRF(xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=RBF(input_dim=2, lengthscale=torch.tensor(0.2), variance=torch.tensor(25.0)),
                                  num_observation_categories=12, num_topic_categories=3, dirichlet_param=0.01, n_points=[4, 3],
                                  fixed_inducing_points=True, inducing_init="grid", maxjitter=15, jitter=1e-6, device=device,
                                  dtype=torch.float64)
    sc = poutine.scale(scale=1.0 / m.N)
    svi = SVI(model=sc(model.model), guide=sc(model.guide), optim=Adam({"lr": 1e-3}), loss=Trace_ELBO())
    g = torch.Generator().manual_seed(2)
    for rows in ([0, 1, 2, 3], [100, 150, 17, 60]):
        eps = torch.randn(3, 4, generator=g, dtype=torch.float64)
        xb, wb = xs[rows].clone(), ws[rows].clone()           # same shape, new storage (possibly the block just freed)
        got = svi.evaluate_loss(xs=xb, ws=wb, eps=eps)
        m.force_jitter_level = model._engine.last_jitter_level
        ref = float(m.loss(eps, xs=m.xs[rows], ws=m.ws[rows], n_global=m.N).detach())
        assert abs(got - ref) <= LOSS_TOL_VS_TORCH * abs(ref), (rows, got, ref, model._engine.read_out(), m.last_terms)
        del xb, wb


def test_clipped_adam_decays_lr_before_the_update():
    """pyro's ClippedAdam multiplies lr by lrd BEFORE forming step_size: three steps with lrd = 0.9."""
    from gdrf_amd.optim import ClippedAdam
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6, optimizer="clippedadam", lr=1e-2)
    for name, p in m.params.items():
        m._get_opt(name, p).lrd = 0.9
    eng = engine_from_oracle(m)
    opt = ClippedAdam({"lr": 1e-2, "lrd": 0.9})
    opt._bind(eng)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        eps = torch.randn(m.K, m.N, generator=g, dtype=torch.float64)
        m.step(eps)
        eng.loss_and_grads(xs, ws, dev(eps, eng))
        opt._step()
    assert abs(opt.lr - 1e-2 * 0.9 ** 3) < 1e-15
    for name in eng.PARAM_NAMES:
        assert relerr(eng.view(name).cpu().numpy(), m.params[name].detach().numpy()) < 1e-8, name


def test_optimizer_state_round_trip_carries_the_inducing_inputs():
    """get_state/set_state (train_script.py:348,495) cover every learnt block, also the ones only some configurations carry."""
    from gdrf_amd.optim import Adam
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6, learn_inducing=True, random_inducing=True, lr=1e-2)
    eng = engine_from_oracle(m)
    opt = Adam({"lr": 1e-2}); opt._bind(eng)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    g = torch.Generator().manual_seed(5)
    for _ in range(2):
        eng.loss_and_grads(xs, ws, dev(torch.randn(m.K, m.N, generator=g, dtype=torch.float64), eng)); opt._step()
    st = opt.get_state()
    assert "inducing_unc" in st and float(st["inducing_unc"]["exp_avg_sq"].abs().max()) > 0
    eng2 = engine_from_oracle(m)
    eng2.params.copy_(eng.params)
    opt2 = Adam({"lr": 1e-2}); opt2._bind(eng2); opt2.set_state(st)
    eps = dev(torch.randn(m.K, m.N, generator=g, dtype=torch.float64), eng)
    for e, o in ((eng, opt), (eng2, opt2)):
        e.loss_and_grads(xs, ws, eps); o._step()
    assert torch.equal(eng.params, eng2.params)


def test_more_than_32_topics():
    """num_topics is unbounded in the reference (train_script.py:111); K = 40 runs the generic row kernels."""
    m, eps = make_oracle(dtype=torch.float64, jitter=1e-6, W=15, H=9, V=11, K=40, n_points=(4, 3))
    eng = engine_from_oracle(m)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    eng.loss_and_grads(xs, ws, dev(eps, eng))
    out = eng.read_out()
    m.force_jitter_level = eng.last_jitter_level
    loss_ref, grads_ref = m.loss_and_grads(eps)
    assert abs(out["loss"] - loss_ref) <= LOSS_TOL_VS_TORCH * abs(loss_ref)
    gv = eng.named_views(eng.grads)
    for name in eng.PARAM_NAMES:
        assert relerr(gv[name].cpu().numpy(), grads_ref[name].numpy()) < 1e-7, name
    assert relerr(eng.predict(xs, 1).cpu().numpy(), m.topic_probs().numpy()) < 1e-9
    assert relerr(eng.predict(xs, 2).cpu().numpy(), m.word_probs().numpy()) < 1e-9
    assert relerr(eng.predict(xs, 0).cpu().numpy(), m.log_topic_probs().numpy()) < 1e-9
    s = eng.predict(xs, 3, ws).cpu().numpy()
    assert abs(float(np.exp(-s[0] / s[1])) - float(m.perplexity())) < 1e-8 * float(m.perplexity())


def test_large_vocabulary_shrinks_the_row_block_and_oversize_fails_loudly():
    m, eps = make_oracle(dtype=torch.float64, jitter=1e-6, W=12, H=9, V=300, K=5, n_points=(4, 3))
    eng = engine_from_oracle(m)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    eng.loss_and_grads(xs, ws, dev(eps, eng))
    out = eng.read_out()
    m.force_jitter_level = eng.last_jitter_level
    loss_ref, grads_ref = m.loss_and_grads(eps)
    assert abs(out["loss"] - loss_ref) <= LOSS_TOL_VS_TORCH * abs(loss_ref)
    assert relerr(eng.named_views(eng.grads)["phi_unc"].cpu().numpy(), grads_ref["phi_unc"].numpy()) < 1e-7
    m2, eps2 = make_oracle(dtype=torch.float64, jitter=1e-6, W=6, H=5, V=2500, K=5, n_points=(3, 2))
    eng2 = engine_from_oracle(m2)
    with pytest.raises(_lib.GdrfHipError, match="too large"):
        eng2.loss_and_grads(dev(m2.xs, eng2), dev(m2.ws, eng2, torch.int32), dev(eps2, eng2))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_packed_payload_sums_like_float64(dtype):
    """The doubles of red_d travel in the tail of the flat payload (one all-reduce per step): emulate a 4-rank sum of the
    packed tails in the payload's element type and compare with the float64 sum."""
    m, _ = make_oracle(dtype=dtype, jitter=1e-4)
    eng = engine_from_oracle(m)
    lib, s = eng.lib, torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    nd = eng.red_d.numel()
    mag = 10.0 ** torch.randint(-3, 9, (nd,), generator=g).double()      # every entry has its own magnitude, shared by the ranks
    vals = [((1.0 + 0.5 * torch.rand(nd, generator=g, dtype=torch.float64)) * torch.randn(nd, generator=g, dtype=torch.float64).sign() * mag)
            .to(eng.device) for _ in range(4)]
    acc = torch.zeros_like(eng.red_T)
    for v in vals:
        eng.red_T.zero_(); eng.red_d.copy_(v)
        _lib.check(lib.gdrf_payload_pack(eng.ctx, eng.red_T.data_ptr(), eng.red_d.data_ptr(), s), "pack")
        acc += eng.red_T                                   # what the all-reduce does, in the payload's dtype
    eng.red_d.zero_()
    _lib.check(lib.gdrf_payload_unpack(eng.ctx, acc.data_ptr(), eng.red_d.data_ptr(), s), "unpack")
    ref = torch.stack(vals).sum(0)
    scale = torch.stack(vals).abs().max(0).values
    err = ((eng.red_d - ref).abs() / scale).max().item()
    assert err < (1e-15 if dtype == torch.float64 else 1e-12), err


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_hyper_gradients_through_the_derivative_product(dtype, kind, monkeypatch):
    """GDRF_WD_PATH=1: the K_nm parts of the lengthscale / variance gradients as sum Wbar o W' and sum Wbar o W with
    W' = (dK_nm / d log ls) Linv^T (a second forward-shaped GEMM) instead of the backward GEMM Kbar = Wbar Linv."""
    monkeypatch.setenv("GDRF_WD_PATH", "1")
    m, eps = make_oracle(dtype=torch.float64, jitter=1e-4, kind=kind, W=23, H=11, V=7, K=3, n_points=(6, 6), lengthscale=0.15)
    loss, grads = m.loss_and_grads(eps)
    eng = engine_from_oracle(m, dtype=dtype)
    eng.loss_and_grads(dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=m.last_jitter_level)
    tol = 1e-8 if dtype == torch.float64 else 2e-3
    for name in ("log_lengthscale", "log_variance"):
        got, ref = float(eng.view(name, eng.grads)), float(grads[name])
        assert abs(got - ref) <= tol * max(abs(ref), 1e-3), (name, got, ref)


# ---- non-unit worlds: the reference's guide scales its inputs twice (quirk Q3, gdrf/models/sparse_gdrf.py:376-380) -------------------
WORLD = [(2.0, 5.0), (-1.0, 3.0)]


def _world_oracle(dtype, guide_rescale=True, mean_function=None, seed=4):
    from oracle.gdrf_oracle import RefShapedGDRF
    from gdrf_amd.data import synth_circles
    xs, ws, _ = synth_circles(17, 11, 9, 3, seed=seed)
    lower = torch.tensor([w[0] for w in WORLD], dtype=torch.float64)
    delta = torch.tensor([w[1] - w[0] for w in WORLD], dtype=torch.float64)
    xs_w = torch.from_numpy(xs).double() * delta + lower                       # observations in world coordinates
    m = RefShapedGDRF(xs_w, ws, kind="rbf", K=3, n_points=(5, 4), lengthscale=0.3, dtype=dtype, jitter=1e-6, world=WORLD,
                      guide_rescale=guide_rescale, optimizer="adam", lr=1e-2, mean_function=mean_function)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        m.params["u_loc"].add_(0.3 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64).to(dtype))
        m.params["phi_unc"].add_(0.5 * torch.randn(m.params["phi_unc"].shape, generator=g, dtype=torch.float64).to(dtype))
        m.params["u_scale_tril_unc"].add_(0.1 * torch.randn(m.params["u_scale_tril_unc"].shape, generator=g, dtype=torch.float64).tril().to(dtype))
    eps = torch.randn(3, m.N, generator=g, dtype=torch.float64).to(dtype)
    return m, eps, xs_w


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("with_mean", [False, True])
def test_non_unit_world_guide_scales_twice_like_the_reference(dtype, with_mean):
    """Loss and every gradient of the two-point evaluation (gdrf_step_local2: guide at scale(scale(xs)), model at scale(xs)) against
    autograd through the reference-shaped oracle with the same double scaling."""
    mf = (lambda x: 1.5 * x[:, 0] - 0.7 * x[:, 1]) if with_mean else None
    m, eps, xs_w = _world_oracle(torch.float64, mean_function=mf)
    loss_ref, grads_ref = m.loss_and_grads(eps)
    eng = engine_from_oracle(m, dtype=dtype)
    xs_m = m.scale(xs_w)
    xs_g = m.scale(xs_m)
    kw = {}
    if with_mean:
        kw = dict(mean=dev(mf(xs_m), eng), mean_guide=dev(mf(xs_g), eng))
    eng.loss_and_grads(dev(xs_m, eng), dev(m.ws, eng, torch.int32), dev(eps, eng), xs_guide=dev(xs_g, eng), force_level=m.last_jitter_level, **kw)
    out = eng.read_out()
    tl, tg = (1e-9, 1e-7) if dtype == torch.float64 else (2e-5, 2e-2)
    assert abs(out["loss"] - loss_ref) <= max(tl, LOSS_TOL_VS_TORCH) * abs(loss_ref), (out["loss"], loss_ref)
    gv = eng.named_views(eng.grads)
    for name in eng.PARAM_NAMES:
        assert relerr(gv[name].cpu().numpy(), grads_ref[name].numpy()) < tg, name
    # it IS a different objective from scaling once on both sides
    m1, _, _ = _world_oracle(torch.float64, guide_rescale=False, mean_function=mf)
    assert abs(float(m1.loss(eps).detach()) - loss_ref) > 1e-3 * abs(loss_ref)


def test_non_unit_world_through_the_model_surface():
    """SparseMultinomialGDRF(world=...) + SVI.step: three Adam steps follow the oracle with the reference's double scaling
    (default), and guide_rescale=False reproduces the unit-cube model on the scaled inputs."""
    from gdrf_amd import poutine
    from gdrf_amd.infer import SVI, Trace_ELBO
    from gdrf_amd.kernels import RBF
    from gdrf_amd.models import SparseMultinomialGDRF
    from gdrf_amd.optim import Adam
    device = "cuda:0"
    for rescale in (True, False):
        m, eps0, xs_w = _world_oracle(torch.float64, guide_rescale=rescale)
        model = SparseMultinomialGDRF(xs=xs_w.to(device), ws=m.ws.to(device), world=WORLD,
                                      kernel=RBF(input_dim=2, lengthscale=torch.tensor(0.3), variance=torch.tensor(25.0)),
                                      num_observation_categories=9, num_topic_categories=3, dirichlet_param=0.01, n_points=[5, 4],
                                      fixed_inducing_points=True, inducing_init="grid", maxjitter=15, jitter=1e-6, device=device,
                                      dtype=torch.float64, guide_rescale=rescale)
        eng = model._engine_for(m.N)
        for name in eng.PARAM_NAMES:
            eng.view(name).copy_(m.params[name].detach().to(eng.device))
        sc = poutine.scale(scale=1.0 / m.N)
        svi = SVI(model=sc(model.model), guide=sc(model.guide), optim=Adam({"lr": 1e-2}), loss=Trace_ELBO())
        g = torch.Generator().manual_seed(9)
        for step in range(3):
            eps = torch.randn(3, m.N, generator=g, dtype=torch.float64)
            got = svi.step(xs=xs_w.to(device), ws=m.ws.to(device), eps=eps)
            ref = m.step(eps)
            assert abs(got - ref) <= LOSS_TOL_VS_TORCH * abs(ref), (rescale, step, got, ref)
        # the predictive path scales once in the reference too (sparse_gdrf.py:161-162)
        assert relerr(model.topic_probs(xs_w.to(device)).cpu().numpy(), m.topic_probs(xs_w).numpy()) < 1e-7
    with pytest.raises(AssertionError):
        model.topic_probs(xs_w.to(device) + 10.0)              # outside the world: _check_bounds (topic_model.py:191-198)


@pytest.mark.parametrize("lengthscale", [0.1, 0.004])
def test_in_step_knm_in_the_solve_precision_matches_numpy_exp(lengthscale):
    """The step's own K_nm (float64 beside float32 arrays; RBF: a straight-line 2^t polynomial instead of the library exp) against
    numpy's exp at the same float32 inputs: round-off only, and exact zeros where the covariance underflows (lengthscale 0.004
    puts far pairs beyond 2^-1100)."""
    m, eps = make_oracle(kind="rbf", W=37, H=11, n_points=(6, 5), dtype=torch.float32, lengthscale=lengthscale)
    eng = engine_from_oracle(m)
    eng.loss_and_grads(dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=3)
    torch.cuda.synchronize()
    got = eng.workspace("Knm", n_rows=m.xs.shape[0]).cpu().numpy()
    assert got.dtype == np.float64
    x, z = m.xs.float().double().numpy(), eng.Z.cpu().double().numpy()
    ls = float(np.exp(np.float64(eng.view("log_lengthscale").cpu().numpy().reshape(-1)[0])))
    var = float(np.exp(np.float64(eng.view("log_variance").cpu().numpy().reshape(-1)[0])))
    r2 = ((x[:, None, :] - z[None, :, :]) ** 2).sum(-1) / ls ** 2
    ref = var * np.exp(-0.5 * r2)
    big = ref > 1e-280
    assert np.abs(got[big] / ref[big] - 1.0).max() < 2e-12 * max(1.0, float(r2[big].max()) / 50)       # the argument's own rounding grows with r2
    assert np.abs(got[~big]).max(initial=0.0) <= 1e-279
    if lengthscale < 0.01:
        assert (ref == 0).any() and (got[ref == 0] == 0).all()


@pytest.mark.parametrize("npts", [(32, 16), (13, 11)])
def test_panelwise_cholesky_backward_error_and_block_inverse(npts):
    """The factorisation on the step's critical path (one launch per 32-column panel, pivots from a v_rsq_f64 seed) at the headline's
    M = 512 and at a ragged M = 143: L L^T reproduces K_uu to double round-off (a pivot only good to 2^-45 would show as 1e-13),
    L matches numpy's factor of the same matrix, and Linv L = 1."""
    m, _ = make_oracle(kind="rbf", W=16, H=9, n_points=npts, dtype=torch.float64, jitter=1e-6, lengthscale=0.1)
    eng = engine_from_oracle(m)
    eng.factorize()
    K = eng.workspace("Kuu").cpu().numpy()
    L = eng.workspace("L").cpu().numpy()
    Li = eng.workspace("Linv").cpu().numpy()
    assert np.abs(np.triu(L, 1)).max() == 0.0
    res = np.abs(L @ L.T - K).max() / np.abs(K).max()
    ref = np.linalg.cholesky(K)
    res_ref = np.abs(ref @ ref.T - K).max() / np.abs(K).max()
    print(f"M={m.M}: backward error {res:.2e} (numpy {res_ref:.2e}), |L - numpy| / |L| {np.abs(L - ref).max() / np.abs(ref).max():.2e}")
    assert res < 4e-15 and res < 8 * res_ref + 1e-16
    assert np.abs(Li @ L - np.eye(m.M)).max() < 1e-7                    # forward error of the inverse: condition-number bound, not round-off
    assert relerr(L, ref) < 1e-8


def _oracle_3d(dtype, kind="rbf", learn_inducing=False, seed=3, n=420, V=11, K=3, n_points=(4, 3, 3)):
    """Spatiotemporal inputs (x, y, t) in the unit cube: the D = 3 shape of the reference's data (index columns of the CSV)."""
    g = torch.Generator().manual_seed(seed)
    xs = torch.rand(n, 3, generator=g, dtype=torch.float64)
    ws = torch.randint(0, 9, (n, V), generator=g, dtype=torch.int32)
    Z = None
    if learn_inducing:
        M = int(np.prod(n_points))
        Z = 0.05 + 0.9 * torch.rand(M, 3, generator=g, dtype=torch.float64)
    m = RefShapedGDRF3(xs, ws, kind=kind, K=K, n_points=n_points, dtype=dtype, jitter=1e-6 if dtype == torch.float64 else 1e-4, lengthscale=0.3,
                       Z=Z, learn_inducing=learn_inducing, optimizer="adam", lr=1e-2)
    with torch.no_grad():
        m.params["u_loc"].add_(0.3 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64).to(dtype))
        m.params["u_scale_tril_unc"].add_(0.1 * torch.randn(m.params["u_scale_tril_unc"].shape, generator=g, dtype=torch.float64).tril().to(dtype))
        m.params["phi_unc"].add_(0.5 * torch.randn(m.params["phi_unc"].shape, generator=g, dtype=torch.float64).to(dtype))
    eps = torch.randn(K, n, generator=g, dtype=torch.float64).to(dtype)
    return m, eps


from oracle.gdrf_oracle import RefShapedGDRF as RefShapedGDRF3  # noqa: E402


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
@pytest.mark.parametrize("learn_inducing", [False, True])
def test_three_dimensional_inputs_fp64(kind, learn_inducing):
    """D = 3 (x, y, t): the K_nm kernels' generic-dimension instantiations, the backward epilogue's general form (the LDS-transposed
    one is for D <= 2) and the inducing-input gradient in three dimensions: loss and every gradient against autograd, then five Adam
    steps."""
    m, eps = _oracle_3d(torch.float64, kind=kind, learn_inducing=learn_inducing)
    eng = engine_from_oracle(m)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    eng.loss_and_grads(xs, ws, dev(eps, eng))
    out = eng.read_out()
    assert out["chol_failed"] == 0
    m.force_jitter_level = eng.last_jitter_level
    loss_ref, grads_ref = m.loss_and_grads(eps)
    assert abs(out["loss"] - loss_ref) <= LOSS_TOL_VS_TORCH * abs(loss_ref)
    gv = eng.named_views(eng.grads)
    names = list(eng.PARAM_NAMES) + (["inducing_unc"] if learn_inducing else [])
    for name in names:
        assert relerr(gv[name].cpu().double().numpy(), grads_ref[name].double().numpy()) < 1e-7, name
    g = torch.Generator().manual_seed(9)
    for step in range(5):
        e = torch.randn(m.K, m.N, generator=g, dtype=torch.float64)
        loss_ref = m.step(e)
        eng.loss_and_grads(xs, ws, dev(e, eng))
        eng.adam("adam", 1e-2)
        assert abs(eng.read_out()["loss"] - loss_ref) <= LOSS_TOL_VS_TORCH * abs(loss_ref), step
    for name in names:
        assert relerr(eng.view(name).cpu().numpy(), m.params[name].detach().numpy()) < 1e-7, name


def test_three_dimensional_inputs_fp32_and_predictive():
    """The same shape through the float32 arrays + float64 solve configuration (f16x3 contractions), and the predictive path."""
    m, eps = _oracle_3d(torch.float32)
    eng = engine_from_oracle(m)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    eng.loss_and_grads(xs, ws, dev(eps, eng))
    out = eng.read_out()
    m64, _ = _oracle_3d(torch.float64)
    with torch.no_grad():
        for name in m.params:
            m64.params[name].copy_(m.params[name].double())
    m64.jitter = m.jitter
    m64.force_jitter_level = eng.last_jitter_level
    loss_ref, grads_ref = m64.loss_and_grads(eps.double(), xs=m.xs.double())
    assert abs(out["loss"] - loss_ref) <= 1e-4 * abs(loss_ref)
    gv = eng.named_views(eng.grads)
    for name in eng.PARAM_NAMES:
        assert relerr(gv[name].cpu().double().numpy(), grads_ref[name].numpy()) < 2e-2, name
    tp = eng.predict(xs, 1).cpu().double().numpy()                      # topic_probs (n, K)
    with torch.no_grad():
        assert np.abs(tp - m64.topic_probs(m.xs.double()).numpy()).max() < 1e-4


_LINKS = {
    "tempered_softmax": lambda mu: torch.softmax(0.5 * mu, -2),                     # sums to one over the topics
    "sigmoid": lambda mu: 0.05 + torch.sigmoid(0.02 * mu),                          # does not: Multinomial normalises theta^T Phi.  (mu = f_loc +
                                                                                    # f_var eps spans +-1e3 here: a bare sigmoid is 0 for every topic of some rows in float32)
    "smoothed_softmax": lambda mu: (torch.softmax(mu, -2) + 0.1) / (1.0 + 0.1 * mu.shape[-2]),          # a floor under every topic
}


@pytest.mark.parametrize("link", sorted(_LINKS))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_custom_link_function_matches_autograd(link, dtype):
    """The reference's `link_function` argument (abstract_gdrf.py:34-50, sparse_gdrf.py:361): the link and its Jacobian are evaluated with
    torch between the three phases of gdrf_step_local_link.  Loss and every gradient against the oracle with the same callable."""
    from oracle.gdrf_oracle import RefShapedGDRF
    from gdrf_amd.data import synth_circles
    xs, ws, _ = synth_circles(17, 9, 13, 4, seed=4)
    m = RefShapedGDRF(xs, ws, kind="rbf", K=4, n_points=(4, 3), lengthscale=0.2, dtype=torch.float64, jitter=1e-6 if dtype == torch.float64 else 1e-4,
                      link_function=_LINKS[link], optimizer="adam", lr=1e-2)
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        m.params["u_loc"].add_(0.3 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64))
        m.params["u_scale_tril_unc"].add_(0.1 * torch.randn(m.params["u_scale_tril_unc"].shape, generator=g, dtype=torch.float64).tril())
        m.params["phi_unc"].add_(0.5 * torch.randn(m.params["phi_unc"].shape, generator=g, dtype=torch.float64))
        if dtype == torch.float32:
            for p in m.params.values():
                p.copy_(p.float().double())
    eps = torch.randn(4, m.N, generator=g, dtype=torch.float64).to(dtype).double()
    eng = engine_from_oracle(m, dtype=dtype)
    eng.link_function = _LINKS[link]
    xs_d, ws_d = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    eng.loss_and_grads(xs_d, ws_d, dev(eps, eng))
    out = eng.read_out()
    assert out["chol_failed"] == 0
    m.force_jitter_level = eng.last_jitter_level
    loss_ref, grads_ref = m.loss_and_grads(eps, xs=m.xs.to(dtype).double())
    tol_l, tol_g = (LOSS_TOL_VS_TORCH, 1e-7) if dtype == torch.float64 else (1e-4, 2e-2)
    assert abs(out["loss"] - loss_ref) <= tol_l * abs(loss_ref), (out["loss"], loss_ref)
    gv = eng.named_views(eng.grads)
    for name in eng.PARAM_NAMES:
        assert relerr(gv[name].cpu().double().numpy(), grads_ref[name].numpy()) < tol_g, name
    if dtype == torch.float64:                      # and a few optimizer steps stay on the oracle's trajectory
        for step in range(3):
            e = torch.randn(4, m.N, generator=g, dtype=torch.float64)
            loss_ref = m.step(e)
            eng.loss_and_grads(xs_d, ws_d, dev(e, eng))
            eng.adam("adam", 1e-2)
            assert np.isfinite(loss_ref) and abs(eng.read_out()["loss"] - loss_ref) <= LOSS_TOL_VS_TORCH * abs(loss_ref), step
        for name in eng.PARAM_NAMES:
            ref_p = m.params[name].detach().numpy()
            assert np.isfinite(ref_p).all() and relerr(eng.view(name).cpu().numpy(), ref_p) < 1e-7, name


def test_custom_link_function_through_the_model_surface():
    """SparseMultinomialGDRF(link_function=...) + SVI.step, and the predictive helpers of abstract_gdrf.py:113-139 with that link."""
    from gdrf_amd.data import synth_circles
    from gdrf_amd.infer import SVI, Trace_ELBO
    from gdrf_amd.kernels import RBF
    from gdrf_amd.models import SparseMultinomialGDRF
    from gdrf_amd.optim import Adam
    from gdrf_amd import poutine
    from oracle.gdrf_oracle import RefShapedGDRF
    link = _LINKS["tempered_softmax"]
    xs_np, ws_np, _ = synth_circles(14, 10, 9, 3, seed=8)
    xs = torch.from_numpy(xs_np).to("cuda:0", torch.float64); ws = torch.from_numpy(ws_np).to("cuda:0")
    model = SparseMultinomialGDRF(xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=RBF(input_dim=2, lengthscale=torch.tensor(0.2), variance=torch.tensor(25.0)),
                                  num_observation_categories=9, num_topic_categories=3, dirichlet_param=0.01, n_points=[4, 3],
                                  fixed_inducing_points=True, inducing_init="grid", maxjitter=15, jitter=1e-6, device="cuda:0",
                                  dtype=torch.float64, seed=3, link_function=link)
    scale = poutine.scale(scale=1.0 / xs.shape[0])
    svi = SVI(model=scale(model.model), guide=scale(model.guide), optim=Adam({"lr": 1e-2}), loss=Trace_ELBO(num_particles=1))
    ref = RefShapedGDRF(xs_np, ws_np, kind="rbf", K=3, n_points=(4, 3), lengthscale=0.2, dtype=torch.float64, jitter=1e-6, link_function=link,
                        optimizer="adam", lr=1e-2)
    eng = model._engine_for(xs.shape[0])
    g = torch.Generator().manual_seed(31)
    for step in range(3):
        e = torch.randn(3, xs.shape[0], generator=g, dtype=torch.float64)
        loss = svi.step(xs=xs, ws=ws, subsample=False, eps=e)
        ref.force_jitter_level = eng.last_jitter_level
        loss_ref = ref.step(e)
        assert abs(loss - loss_ref) <= LOSS_TOL_VS_TORCH * abs(loss_ref), (step, loss, loss_ref)
    with torch.no_grad():
        tp = model.topic_probs(xs).cpu().numpy()
        assert tp.shape == (xs.shape[0], 3)
        assert np.abs(tp - ref.topic_probs(torch.from_numpy(xs_np).double()).numpy()).max() < 1e-8
        ppx = float(model.perplexity(xs, ws))
        assert abs(ppx - float(ref.perplexity(torch.from_numpy(xs_np).double(), torch.from_numpy(ws_np)))) < 1e-6 * ppx


@pytest.mark.parametrize("shape", [(20, 15, (16, 12)), (8, 8, (32, 16))])
def test_wbar_reduction_slices_for_few_rows(shape):
    """Mini-batch sizes: bwd_wbar_f16_k64_kernel splits its reduction blocks over gridDim.y slices into slabs (a handful of workgroups would
    otherwise walk all K x Mp / 64 chunks serially) and wbar_slab_sum_kernel adds them and takes the maximum that scales the G^T operand.
    Wbar and G^T against fp64 products of the engine's own float32 inputs; bit-identical on a second call."""
    W_, H_, npts = shape
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-4, kind="rbf", W=W_, H=H_, V=12, K=5, n_points=npts, lengthscale=0.1)
    eng = engine_from_oracle(m, mfma_mode="f16x3")
    xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
    eng.loss_and_grads(xs, ws, e)
    n = m.N
    Wm = eng.workspace("W", n).cpu().double().numpy()
    vbar = eng.workspace("vbar", n).cpu().double().numpy()
    locbar = eng.workspace("locbar", n).cpu().double().numpy()
    asum = eng.workspace("asum", n).cpu().double().numpy()
    S = eng.workspace("S").cpu().double().numpy()
    U = eng.view("u_loc").cpu().double().numpy()
    ref = locbar.T @ U - 2 * asum[:, None] * Wm
    for k in range(m.K):
        ref += (2 * vbar[k])[:, None] * (Wm @ (S[k] @ S[k].T))
    got = eng.workspace("Wbar", n).cpu()
    assert relerr(got.double().numpy(), ref) < 2e-5
    Mp, lay = (m.M + 31) // 32 * 32, eng.red_layout
    GT = eng.red_T[lay["GT"]:lay["GT"] + Mp * Mp].view(Mp, Mp)[:m.M, :m.M].cpu().double().numpy()
    assert relerr(GT, Wm.T @ got.double().numpy()) < 2e-5
    eng.loss_and_grads(xs, ws, e, force_level=eng.last_jitter_level)
    assert torch.equal(eng.workspace("Wbar", n).cpu(), got)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_factorisation_made_ahead_of_the_step_is_checked_against_its_inputs(dtype):
    """adam() starts the next step's factorisation (gdrf_factorize_mode 1); the step reuses it after a device-side comparison of the kernel
    hyper-parameters and inducing inputs it was made from (mode 2).  (a) The trajectory equals the one without this, bit for bit.
    (b) Writing a hyper-parameter between the update and the next step is noticed: the step redoes itself on the current values."""
    m, _ = make_oracle(dtype=dtype, jitter=1e-6 if dtype == torch.float64 else 1e-4, W=24, H=12, V=9, K=3, n_points=(6, 5), lr=1e-2)
    g = torch.Generator().manual_seed(17)
    eps = [torch.randn(m.K, m.N, generator=g, dtype=torch.float64).to(dtype) for _ in range(4)]
    runs = {}
    for ahead in (True, False):
        eng = engine_from_oracle(m)
        eng.prefactorize = ahead
        xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
        losses = []
        for e in eps:
            eng.loss_and_grads(xs, ws, dev(e, eng))
            eng.adam("adam", 1e-2)
            losses.append(eng.read_out()["loss"])
        runs[ahead] = (losses, eng.params.clone())
    assert runs[True][0] == runs[False][0]
    assert torch.equal(runs[True][1], runs[False][1])
    # (b): a hyper-parameter (input of the factorisation) or a variational parameter (input of the transforms made ahead with it)
    for name, delta in (("log_lengthscale", 0.05), ("u_loc", 0.01), ("u_scale_tril_unc", 0.01), ("log_variance", 0.03)):
        outs = {}
        for ahead in (True, False):
            eng = engine_from_oracle(m)
            eng.prefactorize = ahead
            xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
            eng.loss_and_grads(xs, ws, dev(eps[0], eng))
            eng.adam("adam", 1e-2)                               # (ahead: the work for the next step starts here)
            eng.read_out()
            with torch.no_grad():
                eng.view(name).add_(delta)                       # ... and its input changes behind its back
            eng.loss_and_grads(xs, ws, dev(eps[1], eng))
            first = (eng.read_out()["loss"], eng.grads.clone())
            eng.adam("adam", 1e-2)
            eng.loss_and_grads(xs, ws, dev(eps[2], eng))         # and the step after the redone one is an ordinary one again
            outs[ahead] = first + (eng.read_out()["loss"],)
        assert outs[True][0] == outs[False][0], name
        assert torch.equal(outs[True][1], outs[False][1]), name
        assert outs[True][2] == outs[False][2], name

"""GPU: gradient parity of EVERY parameter block at the sizes that matter (M >= 256, several column tiles, K up to 20), and the
edge cases of the Multinomial log-probability (the eps clamp, and `w == 0 & logit == -inf`).

The HIP gradients (through the C ABI, default arithmetic of each dtype: f16x3 contractions + f64 solve for float32 arrays) are
compared with torch autograd through the reference-shaped fp64 oracle (`RefShapedGDRF.loss_and_grads`, restating
gdrf/models/sparse_gdrf.py:323-409 and the SVI step of gdrf/train_script.py:365-371) at IDENTICAL parameter values, the same
eps and the same jitter level, BEFORE any optimizer step.  Error measure per block: max |got - ref| / max |ref| (a scalar block is
its own maximum).  Bounds asserted below are the ones the build holds, with the measured values printed next to them.
"""
import numpy as np
import pytest
import torch

from tests._util import dev, engine_from_oracle, make_oracle

pytestmark = pytest.mark.gpu

BLOCKS = ("log_lengthscale", "log_variance", "log_noise", "u_loc", "phi_unc", "u_scale_tril_unc")


def _fp32_valued(m):
    """Round the fp64 oracle's parameters AND inducing inputs to float32 values in place: a float32 engine then holds exactly the
    same numbers (the reference keeps both in float32, quirk Q7).  The inducing grid matters: 1/31 is not a float32, and at
    cond(K_uu) ~ 1e7 a 6e-8 relative shift of Z moves the gradients by several 1e-3 - an input difference, not arithmetic error."""
    with torch.no_grad():
        for p in m.params.values():
            p.copy_(p.float().double())
        m.Z = m.Z.float().double()
    return m


def _scalar_block(errs_abs, refs):
    """The three scalar hyper-parameters as ONE block: max |got - ref| over them / max |ref| over them.  A single scalar can be the
    small difference of large path terms (at configs[4]'s shape the fp64 HIP and fp64 autograd values of d loss / d log variance
    already differ by 1e8 ulp), so its error relative to ITSELF measures that cancellation, not the kernels."""
    names = ("log_lengthscale", "log_variance", "log_noise")
    return max(errs_abs[n] for n in names) / max(max(abs(refs[n]) for n in names), 1e-300)


def _all_block_errors(m, eps, dtype, mfma_mode="auto", hyper_backward="auto"):
    eng = engine_from_oracle(m, dtype=dtype, mfma_mode=mfma_mode, hyper_backward=hyper_backward)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    lvl = eng.factorize()                                # the level the engine's own (array-precision) probe asks for
    m.force_jitter_level = lvl
    loss_ref, grads_ref = m.loss_and_grads(eps)
    eng.loss_and_grads(xs, ws, dev(eps, eng), force_level=lvl)
    out = eng.read_out()
    assert out["chol_failed"] == 0
    errs = {"loss": abs(out["loss"] - loss_ref) / abs(loss_ref)}
    eabs, refs = {}, {}
    for name in BLOCKS:
        got = eng.view(name, eng.grads).cpu().double().numpy()
        ref = grads_ref[name].double().numpy()
        assert np.isfinite(got).all(), name
        errs[name] = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300))
        eabs[name], refs[name] = float(np.abs(got - ref).max()), float(np.abs(ref).max())
    errs["hyper_scalars"] = _scalar_block(eabs, refs)
    # the resolution float32 leaves for the argument of the softmax link: mu = loc + v eps is STORED with 24 bits, so theta_j / theta_max =
    # exp(mu_j - mu_max) carries a relative error of |mu| 2^-24 whatever computes it (the all-float32 reference included)
    mu_abs = float(eng.workspace("mu", m.N).abs().max())
    errs["mu_resolution"] = mu_abs * 2.0 ** -24
    print("   |ref| of the scalar gradients:", {n: f"{refs[n]:.3e}" for n in ("log_lengthscale", "log_variance", "log_noise")}, "max |mu| %.1f" % mu_abs)
    return lvl, errs, eng


TENSOR_BLOCKS = ("u_loc", "phi_unc", "u_scale_tril_unc")
# torch's Multinomial.log_prob evaluates lgamma(counts) in float32 for int32 counts (as the reference passes them); the HIP path
# evaluates that data-only constant in float64: fp64 LOSSES agree to ~1e-8 relative, gradients to round-off
LOSS_TOL_FP64 = 1e-7
# float32 arrays, parameters of a contracted ("trained") posterior: |mu| stays O(10) and the gradients are held to this bound
FP32_GRAD_TOL = 1e-3


def _assert_fp32(errs, regime):
    """regime "trained": every block within FP32_GRAD_TOL of autograd.  regime "init" (S_k = L_uu + noise, the reference's initial
    u_scale_tril: tt = |S^T w|^2 and with it mu reach cond(K_uu)-sized values, 3.7e4 at the headline grid): the achievable bound
    is the float32 resolution of mu itself; asserted as a small multiple of it, for all three arithmetic modes alike."""
    assert errs["loss"] < 1e-5
    tol = FP32_GRAD_TOL if regime == "trained" else max(FP32_GRAD_TOL, 6.0 * errs["mu_resolution"])
    for name in TENSOR_BLOCKS + ("hyper_scalars",):
        assert errs[name] < tol, (name, tol, errs)


def test_config1_fp64_all_gradient_blocks():
    """configs[1]'s shape (RBF, M = 256, K = 10, V = 50, fp64) at N = 3000."""
    m, eps = make_oracle(kind="rbf", W=60, H=50, V=50, K=10, n_points=(16, 16), dtype=torch.float64, jitter=1e-8, lengthscale=0.1)
    lvl, errs, _ = _all_block_errors(m, eps, torch.float64)
    print("config 1 fp64: level", lvl, {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["loss"] < LOSS_TOL_FP64
    for name in BLOCKS:
        assert errs[name] < 1e-7, (name, errs)


REGIMES = {"trained": dict(trained_scale=0.3, s_perturb=0.02), "init": dict(s_perturb=0.1)}


@pytest.mark.parametrize("regime", list(REGIMES))
def test_config2_m512_fp32_all_gradient_blocks(regime):
    """configs[2]'s shape class (1-D inputs, M = 512 = four column tiles, K = 8, fp32 arrays) at N = 4000."""
    m, eps = make_oracle(kind="rbf", W=4000, H=1, V=50, K=8, n_points=(512,), one_d=True, dtype=torch.float64, jitter=1e-6,
                         lengthscale=0.02, **REGIMES[regime])
    lvl, errs, eng = _all_block_errors(_fp32_valued(m), eps, torch.float32)
    assert eng.mfma_mode == "f16x3"
    print("config 2 fp32 (f16x3)", regime, ": level", lvl, {k: f"{v:.2e}" for k, v in errs.items()})
    _assert_fp32(errs, regime)


@pytest.mark.parametrize("regime", list(REGIMES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["fp32", "fp64"])
def test_config4_matern52_m1024_k20_all_gradient_blocks(dtype, regime):
    """configs[4]'s shape (Matern-5/2, M = 1024 = eight column tiles, K = 20) at N = 2048: the u_scale_tril gradient runs through
    the all-topics A_k kernel at its full tile count, the lengthscale / variance gradients through G^T, bwd_knm and the Cholesky
    backward."""
    m, eps = make_oracle(kind="matern52", W=64, H=32, V=50, K=20, n_points=(32, 32), dtype=torch.float64, jitter=1e-6,
                         lengthscale=0.1, **REGIMES[regime])
    if dtype == torch.float32:
        _fp32_valued(m)
    lvl, errs, eng = _all_block_errors(m, eps, dtype)
    print("config 4", dtype, regime, "mode", eng.mfma_mode, "level", lvl, {k: f"{v:.2e}" for k, v in errs.items()})
    if dtype == torch.float32:
        _assert_fp32(errs, regime)
    else:
        assert errs["loss"] < LOSS_TOL_FP64
        for name in TENSOR_BLOCKS + ("hyper_scalars",):
            assert errs[name] < 1e-7, (name, errs)


@pytest.mark.parametrize("regime", list(REGIMES))
@pytest.mark.parametrize("mode", ["f16x3", "f16x3-tn", "bf16x6", "f32"])
def test_headline_conditioning_all_gradient_blocks(mode, regime):
    """The headline workload's inducing grid and lengthscale (32 x 16, l = 0.1: cond(K_uu + jitter) ~ 1e7, the jitter level
    escalates) at N = 3000, fp32 arrays, in the default f16x3 arithmetic, the same with the opt-in hyper_backward="tn" (K_nm parts of the
    hyper-parameter gradients through Hd = dK^T Wbar instead of the f64 backward GEMM, csrc/hyper_tn.h), and the two other arithmetics."""
    m, eps = make_oracle(kind="rbf", W=60, H=50, V=50, K=10, n_points=(32, 16), dtype=torch.float64, jitter=1e-6, lengthscale=0.1,
                         **REGIMES[regime])
    mode, hb = (("f16x3", "tn") if mode == "f16x3-tn" else (mode, "auto"))
    lvl, errs, eng = _all_block_errors(_fp32_valued(m), eps, torch.float32, mfma_mode=mode, hyper_backward=hb)
    assert eng.mfma_mode == mode
    assert eng.hyper_backward == ("tn" if hb == "tn" else "f64")
    print("headline conditioning", mode, "hyper_backward", eng.hyper_backward, regime, "level", lvl, {k: f"{v:.2e}" for k, v in errs.items()})
    assert lvl >= 1
    _assert_fp32(errs, regime)


# ---- a10: the edge branches of Multinomial.log_prob ------------------------------------------------------------
def _tiny_word_case(dtype):
    """A word-topic matrix in which two words have probability far below finfo(dtype).eps in every topic: rows that observed the rare word (w > 0, p < eps) take the clamp branch - logit = log(eps), gradient
    through the clamp = 0 -, rows that did not take `w == 0`: 0 * log(clamp) = 0 either way."""
    m, eps = make_oracle(kind="rbf", W=24, H=10, V=12, K=3, n_points=(5, 4), dtype=torch.float64, jitter=1e-6, lengthscale=0.15)
    with torch.no_grad():
        m.params["phi_unc"][:, 0] = -60.0          # softmax: ~ e^-60 = 9e-27 << eps of either dtype
        m.params["phi_unc"][:, 1] = -80.0          # ~ 2e-35 (an exact 0 would make the Dirichlet prior term (alpha - 1) log(0) infinite,
        ws = m.ws.clone()                          #  in the reference too: with the clamp `logit == -inf` cannot occur for finite priors)
        ws[:, 1] = 0                               # nobody observed that word: the `w == 0` side, 0 * log(clamp) = 0
        ws[::7, 0] += 3                            # some rows did observe the very rare one: `w > 0 & p < eps`
        ws[1::7, 0] = 0
        m.ws = ws
    return m, eps


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["fp64", "fp32"])
def test_multinomial_eps_clamp_and_zero_count_branches(dtype):
    m, eps = _tiny_word_case(dtype)
    if dtype == torch.float32:
        _fp32_valued(m)
    eng = engine_from_oracle(m, dtype=dtype)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    lvl = eng.factorize()
    m.force_jitter_level = lvl
    loss_ref, grads_ref = m.loss_and_grads(eps)            # fp64 autograd; torch clamps at finfo(float64).eps there
    assert np.isfinite(loss_ref)
    w_rare = float(m.ws[:, 0].sum())
    assert int((m.ws[:, 0] > 0).sum()) > 10 and w_rare > 30
    if dtype == torch.float32:
        # torch clamps the normalised probabilities to finfo(probs.dtype).eps: a float32 evaluation - the reference's
        # and this engine's - sees log(eps_f32) where the fp64 oracle sees log(eps_f64), for every count on the rare word
        loss_ref -= w_rare * (np.log(np.finfo(np.float32).eps) - np.log(np.finfo(np.float64).eps)) / m.N
    eng.loss_and_grads(xs, ws, dev(eps, eng), force_level=lvl)
    out = eng.read_out()
    assert np.isfinite(out["loss"])
    # the clamp is visible in the value: without it the rare word's counts would add w * log(1e-26) instead of w * log(eps)
    unclamped_shift = w_rare * (np.log(1e-26) - np.log(np.finfo(np.float32 if dtype == torch.float32 else np.float64).eps)) / m.N
    assert abs(unclamped_shift) > 1e-2 * abs(loss_ref)
    ltol = LOSS_TOL_FP64 if dtype == torch.float64 else 1e-5
    assert abs(out["loss"] - loss_ref) < ltol * abs(loss_ref), (out["loss"], loss_ref)
    gtol = 1e-7 if dtype == torch.float64 else 1e-3
    eabs, refs = {}, {}
    for name in BLOCKS:
        got = eng.view(name, eng.grads).cpu().double().numpy()
        ref = grads_ref[name].double().numpy()
        assert np.isfinite(got).all(), name
        eabs[name], refs[name] = float(np.abs(got - ref).max()), float(np.abs(ref).max())
    print("a10", dtype, {n: f"{eabs[n] / max(refs[n], 1e-300):.2e}" for n in BLOCKS})
    for name in TENSOR_BLOCKS:
        assert eabs[name] < gtol * refs[name], (name, eabs[name], refs[name])
    assert _scalar_block(eabs, refs) < gtol, (eabs, refs)
    # perplexity over the same counts takes log(word_probs) without a clamp (abstract_gdrf.py:133-139): rows with w > 0 on a
    # word of probability ~1e-26 stay finite; a count on the p == 0 word would be -inf in the reference too (not exercised)
    pp = eng.predict(xs, 3, ws=ws)
    assert torch.isfinite(pp).all()


# ---- predictive path on the matrix cores (csrc/predict.h), (loc, var) and forward() ------------------------------------------
from oracle.gdrf_oracle import conditional  # noqa: E402
from tests._util import relerr  # noqa: E402

PREDICT_CASES = [
    dict(kind="rbf", W=23, H=11, V=9, K=5, n_points=(6, 5)),                          # N = 253 (ragged 16-row groups), M = 30 (M4 = 32)
    dict(kind="rbf", W=37, H=7, V=11, K=20, n_points=(7, 5)),                         # K = 20: two 16-topic column blocks; M = 35 -> M4 = 36 (tail steps)
    dict(kind="matern52", W=19, H=13, V=9, K=3, n_points=(5, 4)),                     # library-exp covariance branch
    dict(kind="rbf", W=25, H=20, V=8, K=4, n_points=(12,), one_d=True, lengthscale=0.2),
    dict(kind="rbf", W=30, H=20, V=11, K=40, n_points=(6, 5)),                        # K > 32: the one-thread-per-row fallback
]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["fp64", "fp32"])
@pytest.mark.parametrize("case", PREDICT_CASES, ids=lambda c: f"{c['kind']}-K{c['K']}-" + "x".join(map(str, c["n_points"])))
def test_predictive_modes_and_loc_var_against_the_oracle(case, dtype):
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6 if dtype == torch.float64 else 1e-4, **case)
    if dtype == torch.float32:
        _fp32_valued(m)
    eng = engine_from_oracle(m, dtype=dtype)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    lvl = eng.factorize()
    m.force_jitter_level = lvl
    tol = 1e-9 if dtype == torch.float64 else 5e-4
    assert relerr(eng.predict(xs, 0).cpu().numpy(), m.log_topic_probs().numpy()) < tol
    assert relerr(eng.predict(xs, 1).cpu().numpy(), m.topic_probs().numpy()) < tol
    assert relerr(eng.predict(xs, 2).cpu().numpy(), m.word_probs().numpy()) < tol
    s = eng.predict(xs, 3, ws).cpu().numpy()
    perp = float(np.exp(-s[0] / s[1]))
    assert abs(perp - float(m.perplexity())) / float(m.perplexity()) < tol
    # mode 4: (f_loc, f_var) of gp.util.conditional(full_cov=False)
    c = m.constrained()
    with torch.no_grad():
        loc_ref, var_ref = conditional(m.kind, m.xs, m.inducing(), c["lengthscale"], c["variance"], c["u_loc"], c["u_scale_tril"],
                                       m._luu(c), c["scale_mixture"], m.whiten)
    lv = eng.predict(xs, 4).cpu().double().numpy()
    assert lv.shape == (2, m.K, m.N)
    assert relerr(lv[0], loc_ref.numpy()) < tol
    assert relerr(lv[1], var_ref.numpy()) < (1e-9 if dtype == torch.float64 else 2e-3)
    # a subset of rows that is not a multiple of 16, through the same context
    sub = eng.predict(xs[:37].contiguous(), 1).cpu().numpy()
    assert relerr(sub, m.topic_probs().numpy()[:37]) < tol


def test_predictive_path_unwhitened_and_three_dimensional_inputs():
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6, W=21, H=13, K=4, n_points=(5, 4), whiten=False)
    eng = engine_from_oracle(m)
    xs = dev(m.xs, eng)
    m.force_jitter_level = eng.factorize()
    assert relerr(eng.predict(xs, 1).cpu().numpy(), m.topic_probs().numpy()) < 1e-9
    c = m.constrained()
    with torch.no_grad():
        loc_ref, var_ref = conditional(m.kind, m.xs, m.inducing(), c["lengthscale"], c["variance"], c["u_loc"], c["u_scale_tril"],
                                       m._luu(c), c["scale_mixture"], False)
    lv = eng.predict(xs, 4).cpu().numpy()
    assert relerr(lv[0], loc_ref.numpy()) < 1e-9 and relerr(lv[1], var_ref.numpy()) < 1e-9
    # D = 3 takes the four-coordinate instantiation
    from oracle.gdrf_oracle import RefShapedGDRF
    g = torch.Generator().manual_seed(11)
    N, K, V = 150, 3, 7
    xs3 = torch.rand(N, 3, generator=g, dtype=torch.float64)
    ws3 = torch.randint(0, 9, (N, V), generator=g, dtype=torch.int32)
    Z3 = 0.05 + 0.9 * torch.rand(22, 3, generator=g, dtype=torch.float64)
    m3 = RefShapedGDRF(xs3, ws3, kind="rbf", K=K, dtype=torch.float64, jitter=1e-6, lengthscale=0.4, Z=Z3)
    with torch.no_grad():
        m3.params["u_loc"].add_(0.3 * torch.randn(K, 22, generator=g, dtype=torch.float64))
    e3 = engine_from_oracle(m3)
    m3.force_jitter_level = e3.factorize()
    assert relerr(e3.predict(dev(xs3, e3), 1).cpu().numpy(), m3.topic_probs().numpy()) < 1e-9


def test_forward_returns_loc_and_var_like_the_reference_surface():
    from tests.test_gpu_surface import _build
    model, svi, _, xs, ws = _build(dtype=torch.float64, K=3, n_points=(5, 4), W=20, H=12)
    for _ in range(3):
        svi.step(xs=xs, ws=ws, subsample=False)
    loc, var = model.forward(xs)
    assert loc.shape == (3, len(xs)) and var.shape == (3, len(xs))
    assert torch.allclose(loc, model.log_topic_probs(xs), rtol=1e-9, atol=1e-11)        # zero_mean: forward's loc IS log_topic_probs
    assert float(var.min()) > 0
    loc2, var2 = model(xs[:50])                                                             # PyroModule call -> forward
    assert torch.allclose(loc2, loc[:, :50], rtol=1e-9, atol=1e-11) and torch.allclose(var2, var[:, :50], rtol=1e-9, atol=1e-11)
    with pytest.raises(NotImplementedError):
        model.forward(xs, full_cov=True)


# ---- checkpointed model -> end-of-run artefacts (gdrf/utils/loggers.py:35-47, train_script.py:490-506) ---------------------------
@pytest.mark.parametrize("fixed", [True, False], ids=["fixedZ", "learnZ"])
def test_checkpoint_snapshot_regenerates_the_artefact_tables(tmp_path, fixed):
    import copy
    from gdrf_amd import poutine
    from gdrf_amd.infer import SVI, Trace_ELBO
    from gdrf_amd.kernels import Matern52
    from gdrf_amd.models import ModelSnapshot, SparseMultinomialGDRF
    from gdrf_amd.optim import Adam
    from gdrf_amd.data import synth_circles
    devs = "cuda:0"
    xs_np, ws_np, _ = synth_circles(24, 15, 12, 4, seed=5)
    xs, ws = torch.from_numpy(xs_np).float().to(devs), torch.from_numpy(ws_np).int().to(devs)
    model = SparseMultinomialGDRF(xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=Matern52(input_dim=2, lengthscale=0.3, variance=9.0),
                                  num_observation_categories=12, num_topic_categories=4, dirichlet_param=0.05, n_points=[5, 4],
                                  fixed_inducing_points=fixed, inducing_init="grid" if fixed else "random", maxjitter=15, jitter=1e-6,
                                  device=devs, seed=9)
    scale = poutine.scale(scale=1.0 / len(xs))
    svi = SVI(model=scale(model.model), guide=scale(model.guide), optim=Adam({"lr": 0.02}), loss=Trace_ELBO(num_particles=1))
    for _ in range(5):
        svi.step(xs=xs, ws=ws, subsample=False)
    art = model.artifacts(xs, ws)
    assert ("inducing_points" in art) == (not fixed)                       # sparse_gdrf.py:156-157
    if not fixed:
        assert art["inducing_points"].shape == (20, 2)
    ckpt = {"epoch": 4, "best_fitness": -1.0, "model": copy.deepcopy(model).half(), "optimizer": svi.optim.get_state()}
    path = tmp_path / "last.pt"
    torch.save(ckpt, path)
    torch.serialization.add_safe_globals([ModelSnapshot])
    loaded = torch.load(path, map_location="cpu", weights_only=True)["model"]
    assert isinstance(loaded, ModelSnapshot) and loaded.dims == 2 and loaded.K == 4 and loaded.V == 12
    # what _artifacts() computes from the loaded object: half-precision parameters, so compare with the live model at fp16 resolution
    tp, wp, wtm = loaded.topic_probs(xs), loaded.word_probs(xs), loaded.word_topic_matrix
    assert tp.shape == (len(xs), 4) and wp.shape == (len(xs), 12) and wtm.shape == (4, 12)
    assert float((tp - model.topic_probs(xs)).abs().max()) < 2e-2
    assert float((wtm.to(devs) - model.word_topic_matrix).abs().max()) < 2e-3
    # full-precision round trip through the plain-dict payload (no class on the allow-list needed)
    snap = copy.deepcopy(model)
    p2 = tmp_path / "payload.pt"
    torch.save(snap.to_payload(), p2)
    back = ModelSnapshot.from_payload(torch.load(p2, map_location="cpu", weights_only=True))
    assert torch.allclose(back.topic_probs(xs), model.topic_probs(xs), atol=1e-6)
    assert abs(float(back.perplexity(xs, ws)) - float(model.perplexity(xs, ws))) < 1e-4 * float(model.perplexity(xs, ws))


# ---- a factorisation made ahead of its step, parameters written IN STREAM (no host synchronisation in between) -------------------
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["fp64", "fp32"])
def test_in_stream_hyperparameter_write_behind_the_update_is_noticed(dtype):
    """adam() queues the next step's factorisation on the side stream (gdrf_factorize_mode 1); the step compares that factorisation's
    inputs with the current ones on the device (mode 2).  The mismatch word has a line of its own that only the compare kernel
    writes: the side stream's clearing of the Cholesky failure flags - queued behind the update, possibly executed AFTER the next
    step's compare kernel when the host never waits - must not erase it.  Here the lengthscale is rewritten by a kernel on the
    stream right behind adam(), with no host synchronisation before the next step, twenty times in a row; every step must equal the
    one computed without any work made ahead."""
    m, _ = make_oracle(dtype=dtype, jitter=1e-6 if dtype == torch.float64 else 1e-4, W=24, H=12, V=9, K=3, n_points=(6, 5), lr=1e-2)
    g = torch.Generator().manual_seed(23)
    eps = [torch.randn(m.K, m.N, generator=g, dtype=torch.float64).to(dtype) for _ in range(20)]
    runs = {}
    for ahead in (True, False):
        eng = engine_from_oracle(m)
        eng.prefactorize = ahead
        xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
        e_dev = [dev(e, eng) for e in eps]
        bump = torch.full((), 0.01, dtype=eng.dtype, device=eng.device)
        losses = torch.zeros(len(eps), dtype=torch.float64, device=eng.device)
        torch.cuda.synchronize()
        for i, e in enumerate(e_dev):
            eng.loss_and_grads(xs, ws, e)
            losses[i] = eng.out_d[0]                              # device-side copy: the host does not read the loss
            eng.adam("adam", 1e-2)                                # (ahead: the next factorisation starts on the side stream here)
            eng.view("log_lengthscale").add_(bump if i % 2 == 0 else -bump)     # in-stream write of one of its inputs
        runs[ahead] = (losses.cpu(), eng.params.clone())
    assert torch.equal(runs[True][0], runs[False][0])
    assert torch.equal(runs[True][1], runs[False][1])


# ---- bench.py's N > 1 path on one GPU (the driver launches it with torch.distributed.run over RCCL on a multi-GPU node) -----------
@pytest.mark.timeout(900)
def test_bench_multi_rank_path_on_one_gpu_matches_a_single_rank(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` with both ranks on cuda:0 and gloo in place of RCCL
    (GDRF_BENCH_ONE_GPU / GDRF_BENCH_BACKEND: RCCL refuses two ranks on one device): the sharded run must print ONE JSON line with the
    contract's keys, and - the noise being keyed by the global row - end on the single-rank run's loss."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--rows", "30000", "--n-points", "8", "8", "--steps", "3", "--warmup", "1", "--cpu-baseline-n", "0", "--knm-iters", "2",
              "--kernel-pass-steps", "1"]
    env = dict(os.environ, GDRF_BENCH_ONE_GPU="1", GDRF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 31000 + (os.getpid() % 2000)
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                        capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    lines2 = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert len(lines2) == 1, r2.stdout[-2000:]
    d2 = json.loads(lines2[0])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                        env=dict(os.environ), cwd=root, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    for d, n in ((d1, 1), (d2, 2)):
        assert d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "steps/s" and d["higher_is_better"] is True
        assert d["scaling"] == "strong" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
        assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3
        assert d["config"]["N"] == 30000 and d["config"]["rows_per_rank"] == 30000 // n
        assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "cpu_baseline" not in d2                                  # rank 0 at N = 1 only
    # same global rows, same Philox noise, same parameters on every rank: the trajectories agree to the reduction order of the payload
    assert abs(d1["final_loss"] - d2["final_loss"]) < 2e-5 * abs(d1["final_loss"]), (d1["final_loss"], d2["final_loss"])


def test_f16x3_is_rejected_with_the_all_fp32_solve():
    """f16x3 scales W by the bound |w| <= sqrt(variance) that only the f64 solve guarantees: with GDRF_F32_PURE the explicit choice
    fails loudly (auto picks bf16x6 there) instead of overflowing the fp16 pieces."""
    from gdrf_amd import _lib
    from gdrf_amd.engine import Engine
    with pytest.raises(_lib.GdrfHipError, match="f16x3 needs the f64 solve"):
        Engine(256, 12, 3, 9, 2, dtype=torch.float32, pure_fp32=True, mfma_mode="f16x3", process_group=None)
    assert Engine(256, 12, 3, 9, 2, dtype=torch.float32, pure_fp32=True, process_group=None).mfma_mode == "bf16x6"
    assert Engine(256, 12, 3, 9, 2, dtype=torch.float32, process_group=None).mfma_mode == "f16x3"


def test_packed_payload_with_rank_dependent_magnitudes():
    """The doubles of red_d ride in the float payload as four float pieces each (gdrf_payload_pack).  Their sums over the ranks are
    exact while the ranks' values of an entry have similar magnitude (tests/test_gpu_round2.py); with spatially sharded rows the
    inducing-input sums red_d[8..] can differ by orders of magnitude between ranks, and cancel: the float sum of the leading pieces
    then rounds at 2^-24 of the LARGEST summand.  That - float32 resolution of the largest contribution, which is what the float32
    gradient it feeds resolves - is the bound asserted here (float64 contexts copy the doubles and stay exact)."""
    from gdrf_amd import _lib
    m, _ = make_oracle(dtype=torch.float32, jitter=1e-4)
    eng = engine_from_oracle(m)
    lib, s = eng.lib, torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(31)
    nd = eng.red_d.numel()
    vals = []
    for r in range(8):
        mag = 10.0 ** (12.0 * torch.rand(nd, generator=g, dtype=torch.float64) - 4.0)       # every rank its own magnitude per entry: 1e-4 .. 1e8
        vals.append((torch.randn(nd, generator=g, dtype=torch.float64) * mag).to(eng.device))
    vals[1] = -vals[0] * (1.0 + 1e-9)                                                           # and a pair that cancels to 9 digits
    acc = torch.zeros_like(eng.red_T)
    for v in vals:
        eng.red_T.zero_(); eng.red_d.copy_(v)
        _lib.check(lib.gdrf_payload_pack(eng.ctx, eng.red_T.data_ptr(), eng.red_d.data_ptr(), s), "pack")
        acc += eng.red_T
    eng.red_d.zero_()
    _lib.check(lib.gdrf_payload_unpack(eng.ctx, acc.data_ptr(), eng.red_d.data_ptr(), s), "unpack")
    ref = torch.stack(vals).sum(0)
    largest = torch.stack(vals).abs().max(0).values
    err = ((eng.red_d - ref).abs() / largest).max().item()
    print("packed payload, rank-dependent magnitudes: max error / largest summand = %.2e" % err)
    assert err < 2.0 ** -22, err

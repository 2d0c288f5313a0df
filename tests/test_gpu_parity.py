"""GPU parity tests: every stage of the HIP path (through the C ABI) against the CPU oracle.

fp64: tight tolerances (the HIP path and the oracle compute the same real-number function; the
only differences are the direct (x-z)^2 kernel form and summation order).  fp32: compared with
the fp64 oracle evaluated at the same fp32 parameters and the same jitter level.
"""
import numpy as np
import pytest
import torch

from tests._util import dev, engine_from_oracle, make_oracle, relerr, load_params
from oracle.gdrf_oracle import fused_elbo_and_grads, jitter_total, _np_kernel

pytestmark = pytest.mark.gpu

TOL = {torch.float64: dict(k=1e-12, mm=1e-9, w=1e-8, g=1e-7, loss=1e-10),
       torch.float32: dict(k=2e-6, mm=2e-3, w=2e-3, g=2e-2, loss=1e-4)}
# torch's Multinomial.log_prob evaluates lgamma(counts) in float32 when the counts are int32 (as the
# reference passes them, train_script.py:268); the HIP path evaluates that data-only constant in
# float64.  The two losses therefore agree to ~1e-7 relative, not to fp64 round-off.
LOSS_TOL_VS_TORCH = 1e-6


def _aux(m, eps, level):
    P = {k: v.detach().double().numpy().copy() for k, v in m.params.items()}
    return fused_elbo_and_grads(m.kind, m.xs.double().numpy(), m.ws.numpy(), m.Z.double().numpy(), P,
                                m.alpha.double().numpy(), eps.double().numpy(), jitter_total(m.jitter, level))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind", ["rbf", "matern52", "matern32", "exponential"])
@pytest.mark.parametrize("shape", [(16, 9, (4, 3)), (37, 5, (5, 5)), (300, 1, (40,))])
def test_knm(dtype, kind, shape):
    W, H, npts = shape
    one_d = len(npts) == 1
    m, _ = make_oracle(kind=kind, W=W, H=H, n_points=npts, dtype=dtype, one_d=one_d)
    eng = engine_from_oracle(m)
    out = eng.knm(dev(m.xs, eng)).cpu().double().numpy()
    ref, _ = _np_kernel(kind, m.xs.double().numpy(), m.Z.double().numpy(),
                        float(m.params["log_lengthscale"].detach().exp()), float(m.params["log_variance"].detach().exp()))
    assert out.shape == ref.shape
    assert relerr(out, ref) < TOL[dtype]["k"] * 50


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("npts", [(4, 3), (8, 8), (13, 11), (25,)])
def test_factorize(dtype, npts):
    one_d = len(npts) == 1
    m, eps = make_oracle(n_points=npts, dtype=dtype, one_d=one_d, W=40 if one_d else 16, jitter=1e-4, lengthscale=0.05)
    eng = engine_from_oracle(m)
    lvl = eng.factorize()
    _, _, aux = _aux(m, eps, lvl)
    t = TOL[dtype]
    assert relerr(eng.workspace("Kuu").cpu(), aux["Kuu"]) < t["k"] * 50
    L = eng.workspace("L").cpu().double().numpy()
    assert relerr(L, aux["L"]) < t["mm"]
    Li = eng.workspace("Linv").cpu().double().numpy()
    assert relerr(Li, aux["Linv"]) < t["mm"] * 10
    assert relerr(eng.workspace("LinvT").cpu().double().numpy(), Li.T) == 0.0
    # L L^T = Kuu and Linv L = I in the library's own arithmetic
    assert relerr(L @ L.T, aux["Kuu"]) < t["mm"]
    assert np.abs(Li @ L - np.eye(m.M)).max() < t["mm"] * 10


CASES = [
    dict(kind="rbf", W=16, H=9, V=20, K=4, n_points=(4, 3)),          # N=144, M=12 (padding everywhere)
    dict(kind="matern52", W=23, H=11, V=7, K=3, n_points=(6, 6)),     # N=253 (ragged tiles), M=36
    dict(kind="rbf", W=301, H=1, V=50, K=10, n_points=(160,), one_d=True, lengthscale=0.005),   # M=160: 2 column tiles
    dict(kind="rbf", W=20, H=20, V=12, K=1, n_points=(12, 12), lengthscale=0.05),       # K=1, M=144
    dict(kind="matern32", W=19, H=13, V=9, K=3, n_points=(5, 4)),
    dict(kind="exponential", W=19, H=13, V=9, K=3, n_points=(5, 4)),
]


@pytest.mark.parametrize("store_t", [True, False])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("case", CASES)
def test_step_stages_and_grads(dtype, case, store_t):
    """store_t: both forms of the Wbar contraction (stored T_k, triangular; dense W B_k)."""
    case = dict(case)
    m, eps = make_oracle(dtype=dtype, jitter=1e-6 if dtype == torch.float64 else 1e-4, **case)
    eng = engine_from_oracle(m, store_t=store_t)
    assert eng.stores_t == store_t
    assert not engine_from_oracle(m).stores_t          # default: dense form (measured faster)
    xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
    eng.loss_and_grads(xs, ws, e)
    out = eng.read_out()
    assert out["chol_failed"] == 0
    lvl = eng.last_jitter_level
    m.force_jitter_level = lvl
    loss_ref, grads_ref = m.loss_and_grads(eps)                         # autograd, reference-shaped
    _, g_np, aux = _aux(m, eps, lvl)                                    # fp64 stage values at the same parameters
    t = TOL[dtype]
    n = m.N
    report = {}
    for name, key, tol in [("W", "W", t["w"]), ("q", "q", t["w"]), ("loc", "loc", t["w"]), ("tt", "tt", t["w"]),
                           ("mu", "mu", t["w"]), ("vbar", "vbar", t["w"] * 10), ("locbar", "locbar", t["w"] * 10),
                           ("Wbar", "Wbar", t["w"] * 10)]:
        got = eng.workspace(name, n).cpu().double().numpy()
        report[name] = relerr(got, aux[key])
    gviews = eng.named_views(eng.grads)
    for name in eng.PARAM_NAMES:
        ref = g_np[name] if dtype == torch.float32 else grads_ref[name].double().numpy()
        report["g_" + name] = relerr(gviews[name].cpu().double().numpy(), ref)
    loss_np = float(_aux(m, eps, lvl)[0])
    report["loss"] = abs(out["loss"] - loss_np) / abs(loss_np)
    report["loss_vs_torch"] = abs(out["loss"] - loss_ref) / abs(loss_ref)
    print(dtype, case, "level", lvl, {k: f"{v:.2e}" for k, v in report.items()})
    for name in ["W", "q", "loc", "tt", "mu"]:
        assert report[name] < t["w"], (name, report)
    for name in ["vbar", "locbar", "Wbar"]:
        assert report[name] < t["w"] * 10, (name, report)
    for name in eng.PARAM_NAMES:
        assert report["g_" + name] < t["g"], (name, report)
    assert report["loss"] < t["loss"], report
    assert report["loss_vs_torch"] < max(LOSS_TOL_VS_TORCH, t["loss"]), report


@pytest.mark.parametrize("opt", ["adam", "adamw", "clippedadam"])
def test_five_optimizer_steps_fp64(opt):
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6, optimizer=opt, lr=1e-2)
    eng = engine_from_oracle(m)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    g = torch.Generator().manual_seed(5)
    for step in range(5):
        eps = torch.randn(m.K, m.N, generator=g, dtype=torch.float64)
        loss_ref = m.step(eps)
        eng.loss_and_grads(xs, ws, dev(eps, eng))
        eng.adam(opt, 1e-2, weight_decay=0.01 if opt == "adamw" else 0.0)
        out = eng.read_out()
        assert abs(out["loss"] - loss_ref) / abs(loss_ref) < LOSS_TOL_VS_TORCH, (step, out, loss_ref)
    for name in eng.PARAM_NAMES:
        assert relerr(eng.view(name).cpu().numpy(), m.params[name].detach().numpy()) < 1e-8, name


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_predictive_path(dtype):
    m, _ = make_oracle(dtype=dtype, jitter=1e-6 if dtype == torch.float64 else 1e-4, W=23, H=11, V=9, K=5, n_points=(6, 5))
    eng = engine_from_oracle(m)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    m64, _ = make_oracle(dtype=torch.float64, jitter=m.jitter, W=23, H=11, V=9, K=5, n_points=(6, 5))
    for k in m.params:
        m64.params[k] = m.params[k].detach().double()
    lvl = eng.factorize()
    m64.force_jitter_level = lvl
    tol = 1e-9 if dtype == torch.float64 else 5e-4
    assert relerr(eng.predict(xs, 0).cpu().numpy(), m64.log_topic_probs().numpy()) < tol
    assert relerr(eng.predict(xs, 1).cpu().numpy(), m64.topic_probs().numpy()) < tol
    assert relerr(eng.predict(xs, 2).cpu().numpy(), m64.word_probs().numpy()) < tol
    s = eng.predict(xs, 3, ws).cpu().numpy()
    perp = float(np.exp(-s[0] / s[1]))
    assert abs(perp - float(m64.perplexity())) / float(m64.perplexity()) < tol


def test_jitter_retry_duplicate_inducing_points():
    """a singular K_uu (duplicated inducing point) needs the cumulative jitter schedule."""
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-8, n_points=(4, 3))
    Z = m.Z.clone()
    Z[1] = Z[0]
    m.Z = Z
    eng = engine_from_oracle(m)
    lvl = eng.factorize()
    assert 1 <= lvl < 8
    L = eng.workspace("L").cpu().double().numpy()
    assert np.isfinite(L).all()
    eng.maxjitter = 1
    with pytest.raises(RuntimeError, match="reached max jitter"):
        eng.factorize()


def test_fill_eps_is_sharding_invariant_and_standard_normal():
    m, _ = make_oracle(dtype=torch.float32)
    eng = engine_from_oracle(m, n_cap=200000)
    full = eng.fill_eps(seed=1234, step=7, n_offset=0, n=200000)
    a = eng.fill_eps(seed=1234, step=7, n_offset=0, n=120000)
    b = eng.fill_eps(seed=1234, step=7, n_offset=120000, n=80000)
    assert torch.equal(full, torch.cat([a, b], dim=1))
    other = eng.fill_eps(seed=1234, step=8, n_offset=0, n=200000)
    assert not torch.equal(full, other)
    x = full.double().cpu().numpy().ravel()
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1) < 5e-3
    assert abs((x ** 4).mean() - 3.0) < 0.05


def test_multiple_particles_average_the_estimator():
    """Trace_ELBO(num_particles=P, vectorize_particles=True): loss and gradient are the mean over P noise draws."""
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6)
    eng = engine_from_oracle(m)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    g = torch.Generator().manual_seed(9)
    eps = torch.randn(3, m.K, m.N, generator=g, dtype=torch.float64)
    losses, grads = [], []
    for p in range(3):
        l, gr = m.loss_and_grads(eps[p])
        losses.append(l); grads.append(gr)
    eng.loss_and_grads(xs, ws, dev(eps, eng))
    out = eng.read_out()
    assert abs(out["loss"] - np.mean(losses)) < LOSS_TOL_VS_TORCH * abs(np.mean(losses))
    gv = eng.named_views(eng.grads)
    for name in eng.PARAM_NAMES:
        ref = sum(gr[name] for gr in grads).double().numpy() / 3
        assert relerr(gv[name].cpu().numpy(), ref) < 1e-7, name


SPLIT_MODES = ("f16x3", "bf16x6")


@pytest.mark.parametrize("case", [CASES[1], CASES[2]])
def test_split_wbar_is_f32_accurate(case):
    """mfma_mode="f16x3" / "bf16x6": the Wbar contraction on the 16-bit matrix path with split operands (csrc/gemm_split.h)
    must be as accurate as the native f32 MFMA form: all are compared with the fp64 oracle."""
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-4, **dict(case))
    errs = {}
    for mode in ("f32",) + SPLIT_MODES:
        eng = engine_from_oracle(m, mfma_mode=mode, store_t=False)
        xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
        eng.loss_and_grads(xs, ws, e)
        lvl = eng.last_jitter_level
        _, g_np, aux = _aux(m, eps, lvl)
        wbar = eng.workspace("Wbar", m.N).cpu().double().numpy()
        # compare with the oracle's Wbar evaluated from the engine's own (fp32) vbar/locbar inputs is not possible here;
        # use the end-to-end quantities that depend on Wbar linearly: G^T-driven hyper-parameter gradients and Wbar itself
        errs[mode] = dict(wbar=relerr(wbar, aux["Wbar"]),
                          g_ls=relerr(eng.view("log_lengthscale", eng.grads).cpu().numpy(), g_np["log_lengthscale"]),
                          g_var=relerr(eng.view("log_variance", eng.grads).cpu().numpy(), g_np["log_variance"]))
    print(case["kind"], errs)
    for mode in SPLIT_MODES:
        assert errs[mode]["wbar"] < max(4 * errs["f32"]["wbar"], 2e-6), errs
        # (scalars: relative to themselves, i.e. including their own cancellation - the floor is the native-f32 Wbar error level here)
        assert errs[mode]["g_ls"] < max(4 * errs["f32"]["g_ls"], 5e-4), errs
        assert errs[mode]["g_var"] < max(4 * errs["f32"]["g_var"], 5e-4), errs


def _perturb_wide(m, spread):
    """Give the variational factors a wide dynamic range: S_k rows and u_loc scaled by powers of ten drawn per topic / row -
    the block-scaled fp16 split must hold its accuracy when an operand's entries span many binades."""
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        K, M = m.params["u_loc"].shape
        rowscale = 10.0 ** (spread * (torch.rand(K, M, 1, generator=g) - 0.5))
        S = torch.distributions.transform_to(torch.distributions.constraints.lower_cholesky)(m.params["u_scale_tril_unc"].double())
        S = S * rowscale.double()
        m.params["u_scale_tril_unc"].copy_(torch.distributions.transform_to(torch.distributions.constraints.lower_cholesky).inv(S).to(m.dtype))
        m.params["u_loc"].mul_(10.0 ** (spread * (torch.rand(K, M, generator=g) - 0.5)).to(m.dtype))


@pytest.mark.parametrize("spread", [0.0, 3.0, 7.0])
def test_split_modes_against_fp64_product_of_the_same_inputs(spread):
    """Isolates the GEMM arithmetic: Wbar, tt, A_k and G^T are recomputed in fp64 (numpy) from the engine's OWN fp32 inputs (W,
    vbar, locbar, asum, S, u_loc), so the only difference left is how the kernel multiplies.  Each split form must be as close
    to that fp64 product as the native f32 MFMA form is, and they must agree with each other to f32 rounding.  spread > 0:
    the entries of S_k (hence of B_k) span that many decades inside one block scale."""
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-4, kind="rbf", W=40, H=25, V=20, K=6, n_points=(12, 12), lengthscale=0.08)
    if spread:
        _perturb_wide(m, spread)
    out, tt_err, tn_err = {}, {}, {}
    for mode in ("f32",) + SPLIT_MODES:
        eng = engine_from_oracle(m, mfma_mode=mode, store_t=False)
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
        out[mode] = (eng.workspace("Wbar", n).cpu().double().numpy(), ref)
        tt_ref = np.stack([((Wm @ S[k]) ** 2).sum(1) for k in range(m.K)])
        tt_err[mode] = relerr(eng.workspace("tt", n).cpu().double().numpy(), tt_ref)
        # the reductions over the observations: A_k = W^T diag(vbar_k) W (lower triangle) and GT = W^T Wbar
        Mp, lay = (m.M + 31) // 32 * 32, eng.red_layout
        A = eng.red_T[lay["A"]:lay["A"] + m.K * Mp * Mp].view(m.K, Mp, Mp)[:, :m.M, :m.M].cpu().double().numpy()
        GT = eng.red_T[lay["GT"]:lay["GT"] + Mp * Mp].view(Mp, Mp)[:m.M, :m.M].cpu().double().numpy()
        A_ref = np.stack([Wm.T @ (vbar[k][:, None] * Wm) for k in range(m.K)])
        tn_err[mode] = dict(A=relerr(np.tril(A), np.tril(A_ref)), GT=relerr(GT, Wm.T @ out[mode][0]))
    e_f32 = relerr(*out["f32"])
    # measured on MI355X (round 1): f32 MFMA 3.1e-6, bf16x6 2.2e-6 (relative to max|Wbar|; the sum cancels large terms)
    print("spread", spread, "Wbar vs fp64 product of the same inputs:", {k: "%.2e" % relerr(*v) for k, v in out.items()},
          "| vs f32 MFMA:", {k: "%.2e" % relerr(out[k][0], out["f32"][0]) for k in SPLIT_MODES})
    print("tt = |S_k^T w|^2 vs fp64 product of the same inputs:", tt_err)
    print("A_k, GT vs fp64 products of the same inputs:", tn_err)
    assert tt_err["f32"] < 2e-6 and e_f32 < 2e-5
    for mode in SPLIT_MODES:
        assert tt_err[mode] < 1.5 * tt_err["f32"] + 1e-7, (mode, tt_err)
        for q in ("A", "GT"):
            assert tn_err["f32"][q] < 2e-5 and tn_err[mode][q] < 1.5 * tn_err["f32"][q] + 1e-7, (mode, tn_err)
        if spread >= 6.0 and mode == "f16x3":
            # B_k = S_k S_k^T then spans 2 x spread = 14 decades inside ONE block scale: fp16 pieces keep their 22 bits over the 5.4 decades
            # below the block maximum (8.7 for the row-scaled operands of A_k / G^T, whose low piece is stored up-scaled) and lose the low
            # piece gradually further down.  Measured: Wbar 2.8e-6 of its maximum against 1.8e-7 on the native f32 MFMA - the documented
            # price of the default arithmetic on a posterior whose scale factors differ by seven orders of magnitude between inducing points
            assert relerr(*out[mode]) < 1e-5, (mode, relerr(*out[mode]))
        else:
            assert relerr(*out[mode]) < 1.5 * e_f32 + 1e-7, mode


@pytest.mark.parametrize("mode", SPLIT_MODES)
def test_split_kernels_are_deterministic(mode):
    """Two engines, two calls each: tt and Wbar of the split kernels are bit-identical (fixed accumulation order; the only
    atomics are integer maxima, which are order-independent)."""
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-4, kind="rbf", W=40, H=25, V=20, K=6, n_points=(12, 12), lengthscale=0.08)
    got = []
    for _ in range(2):
        eng = engine_from_oracle(m, mfma_mode=mode, store_t=False)
        xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
        for _ in range(2):
            eng.loss_and_grads(xs, ws, e)
            got.append((eng.workspace("tt", m.N).cpu(), eng.workspace("Wbar", m.N).cpu()))
    for tt, wb in got[1:]:
        assert torch.equal(tt, got[0][0]) and torch.equal(wb, got[0][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_speculative_jitter_level_gives_identical_steps(dtype):
    """Engine.loss_and_grads starts on the previous step's jitter level while the probe runs on a second stream; the
    parameters after several Adam steps must be bit-identical to the synchronous probe-then-factorise order, also when the
    guess is wrong (forced here) and the step is redone."""
    m, eps = make_oracle(dtype=dtype, jitter=1e-6, kind="rbf", W=30, H=20, V=12, K=4, n_points=(8, 6), lengthscale=0.15)
    runs = {}
    for mode in ("sync", "speculate", "wrong_guess"):
        eng = engine_from_oracle(m)
        eng.speculate = mode != "sync"
        xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
        losses, levels = [], []
        for t in range(4):
            if mode == "wrong_guess" and t >= 1:
                eng._guess_level = eng.last_jitter_level + 1
            eng.loss_and_grads(xs, ws, e)
            eng.adam("adam", 1e-2)
            losses.append(eng.read_out()["loss"])
            levels.append(eng.last_jitter_level)
        runs[mode] = (eng.params.clone().cpu(), losses, levels)
    for mode in ("speculate", "wrong_guess"):
        assert torch.equal(runs[mode][0], runs["sync"][0]), mode
        assert runs[mode][1] == runs["sync"][1] and runs[mode][2] == runs["sync"][2], mode


# ---- learnable inducing inputs (fixed_inducing_points=False; gdrf/models/sparse_gdrf.py:79-88) ----------------------------
LZ_CASES = [dict(kind="rbf"), dict(kind="matern52", lengthscale=0.3), dict(kind="matern32", lengthscale=0.3),
            dict(kind="exponential", lengthscale=0.4), dict(kind="rbf", one_d=True, n_points=(10,), lengthscale=0.2)]


@pytest.mark.parametrize("case", LZ_CASES, ids=lambda c: c["kind"] + ("_1d" if c.get("one_d") else ""))
def test_learnable_inducing_gradient_matches_autograd(case):
    """d loss / d unconstrained inducing inputs (K_nm path in bwd_knm, K_uu path and sigmoid Jacobian in step_finish) and every
    other gradient, fp64, against torch autograd through the reference-shaped oracle."""
    kw = dict(W=24, H=15, V=10, K=3, n_points=(5, 4), jitter=1e-6, lengthscale=0.25)
    kw.update(case)
    m, eps = make_oracle(dtype=torch.float64, learn_inducing=True, random_inducing=True, **kw)
    loss, grads = m.loss_and_grads(eps)
    eng = engine_from_oracle(m)
    assert eng.learn_inducing
    eng.loss_and_grads(dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=m.last_jitter_level)
    out = eng.read_out()
    assert abs(out["loss"] - float(loss)) < LOSS_TOL_VS_TORCH * abs(float(loss))
    assert torch.allclose(eng.Z.cpu(), m.inducing().detach(), rtol=0, atol=1e-15)
    gz = eng.view("inducing_unc", eng.grads).cpu().numpy()
    # exponential: 1/r in dk/dr2; 1-D: ten random points on a line give a badly conditioned K_uu, which amplifies the
    # difference between the oracle's expanded-form distance and the direct (x-z)^2 here (measured 5e-7)
    tol = 1e-6 if case["kind"] == "exponential" else (5e-6 if case.get("one_d") else 1e-8)
    assert relerr(gz, grads["inducing_unc"].numpy()) < tol
    for name in eng.PARAM_NAMES:
        got, ref = eng.view(name, eng.grads).cpu().numpy(), grads[name].numpy()
        assert np.abs(got - ref).max() < max(tol, 1e-8) * max(np.abs(ref).max(), 1e-6), name      # some blocks are exactly 0


def test_learnable_inducing_adam_steps_follow_the_oracle():
    """Five Adam steps with Z learnable: parameters (incl. the unconstrained inducing block) track the oracle in fp64, and the
    default fp32 build (f64 solve, bf16x6 contractions) gives the same Z gradient to fp32 accuracy at identical parameters."""
    m, eps = make_oracle(dtype=torch.float64, learn_inducing=True, random_inducing=True, W=24, H=15, V=10, K=3, n_points=(5, 4),
                         jitter=1e-6, lengthscale=0.25, lr=1e-2)
    eng = engine_from_oracle(m)
    xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
    eng32 = engine_from_oracle(m, dtype=torch.float32)
    eng32.loss_and_grads(dev(m.xs, eng32), dev(m.ws, eng32, torch.int32), dev(eps, eng32))
    _, g0 = m.loss_and_grads(eps)
    assert relerr(eng32.view("inducing_unc", eng32.grads).cpu().numpy(), g0["inducing_unc"].numpy()) < 2e-3
    for _ in range(5):
        m.step(eps)
        eng.loss_and_grads(xs, ws, e)
        eng.adam("adam", 1e-2)
    assert relerr(eng.view("inducing_unc").cpu().numpy(), m.params["inducing_unc"].detach().numpy()) < 1e-7
    for name in eng.PARAM_NAMES:
        assert relerr(eng.view(name).cpu().numpy(), m.params[name].detach().numpy()) < 1e-7, name
    z0 = torch.sigmoid(transform_inv_check(m))
    assert not torch.allclose(m.inducing().detach(), z0)        # the inducing inputs did move


def transform_inv_check(m):
    # the oracle's initial unconstrained block is recoverable from its fixed copy of Z
    fi = torch.finfo(m.dtype)
    y = m.Z.clamp(min=fi.tiny, max=1.0 - fi.eps)
    return y.log() - (-y).log1p()


# ---- RationalQuadratic kernel (pyro kernels.isotropic.RationalQuadratic; gdrf/train_script.py:93-99) -----------------------
@pytest.mark.parametrize("learn_z", [False, True])
def test_rationalquadratic_kernel_matches_autograd(learn_z):
    """K_nm, the loss and every gradient (incl. the third hyper-parameter scale_mixture, from both the K_nm and the K_uu
    path) for variance * (1 + r2 / (2 scale_mixture))^(-scale_mixture), fp64 against the oracle's autograd."""
    m, eps = make_oracle(dtype=torch.float64, kind="rationalquadratic", scale_mixture=1.7, W=24, H=15, V=10, K=3,
                         n_points=(5, 4), jitter=1e-6, lengthscale=0.25, learn_inducing=learn_z, random_inducing=learn_z)
    loss, grads = m.loss_and_grads(eps)
    eng = engine_from_oracle(m)
    xs = dev(m.xs, eng)
    c = m.constrained()
    from oracle.gdrf_oracle import kernel_matrix
    ref_k = kernel_matrix(m.kind, m.xs, m.inducing().detach(), c["lengthscale"], c["variance"], c["scale_mixture"]).detach().numpy()
    eng.refresh_inducing()
    assert relerr(eng.knm(xs).cpu().numpy(), ref_k) < 1e-12
    eng.loss_and_grads(xs, dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=m.last_jitter_level)
    assert abs(eng.read_out()["loss"] - float(loss)) < LOSS_TOL_VS_TORCH * abs(float(loss))
    assert set(eng.param_names) == set(m.params)
    for name in eng.param_names:
        got, ref = eng.view(name, eng.grads).cpu().numpy(), grads[name].numpy()
        assert np.abs(got - ref).max() < 1e-8 * max(np.abs(ref).max(), 1e-6), name        # some blocks are ~0 (1e-11)
    # and the default fp32 build at the same parameters
    e32 = engine_from_oracle(m, dtype=torch.float32)
    e32.loss_and_grads(dev(m.xs, e32), dev(m.ws, e32, torch.int32), dev(eps, e32))
    g32 = float(e32.view("log_scale_mixture", e32.grads).cpu())
    assert abs(g32 - float(grads["log_scale_mixture"])) < 5e-3 * abs(float(grads["log_scale_mixture"])) + 1e-9


@pytest.mark.parametrize("alpha", [0.0, 0.5, 2.0])
def test_renyi_elbo_matches_the_oracle(alpha):
    """RenyiELBO(alpha, num_particles=3): importance-weighted combination of the particles' payloads on the device against
    autograd through -(logsumexp((1 - alpha) elbo_p) - log P) / (1 - alpha)."""
    m, _ = make_oracle(dtype=torch.float64, jitter=1e-6)
    eng = engine_from_oracle(m)
    g = torch.Generator().manual_seed(11)
    eps = 3.0 * torch.randn(3, m.K, m.N, generator=g, dtype=torch.float64)       # wide draws: clearly unequal weights
    loss, grads = m.loss_and_grads(eps, renyi_alpha=alpha)
    single = [m.loss_and_grads(eps[p])[0] for p in range(3)]
    assert max(single) - min(single) > 1e-3 * abs(loss)
    eng.loss_and_grads(dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng), renyi_alpha=alpha)
    out = eng.read_out()
    assert abs(out["loss"] - loss) < LOSS_TOL_VS_TORCH * abs(loss)
    for name in eng.PARAM_NAMES:
        assert relerr(eng.view(name, eng.grads).cpu().numpy(), grads[name].numpy()) < 1e-7, name


# ---- whiten = False (sparse_gdrf.py:30; the unwhitened branch of pyro's conditional) ---------------------------------------
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_unwhitened_variational_parameters_match_the_oracle(dtype):
    """u' = L^-1 u, S' = L^-1 S in the forward; gradients chained through L^-T and the extra L-dependence into the Cholesky
    backward.  fp64: every gradient and the predictive against autograd; fp32 (default build): against the fp64 oracle at
    the same parameters."""
    m, eps = make_oracle(dtype=torch.float64, whiten=False, W=24, H=15, V=10, K=3, n_points=(5, 4), jitter=1e-6, lengthscale=0.25)
    with torch.no_grad():                       # make u_loc / u_scale_tril generic (not the init L) so that every term is exercised
        g = torch.Generator().manual_seed(3)
        m.params["u_loc"].add_(0.5 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64))
    loss, grads = m.loss_and_grads(eps)
    eng = engine_from_oracle(m, dtype=dtype)
    assert not eng.whiten
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    eng.loss_and_grads(xs, ws, dev(eps, eng), force_level=m.last_jitter_level if dtype == torch.float64 else None)
    out = eng.read_out()
    tol_l, tol_g = (LOSS_TOL_VS_TORCH, 1e-8) if dtype == torch.float64 else (5e-6, 3e-3)
    assert abs(out["loss"] - loss) < tol_l * abs(loss)
    for name in eng.PARAM_NAMES:
        got, ref = eng.view(name, eng.grads).cpu().double().numpy(), grads[name].numpy()
        assert np.abs(got - ref).max() < tol_g * max(np.abs(ref).max(), 1e-6), name
    tp = eng.predict(xs, 1).cpu().double().numpy()
    assert np.abs(tp - m.topic_probs().numpy()).max() < (1e-10 if dtype == torch.float64 else 1e-5)
    # and it is a different model from the whitened one at the same numbers
    m2, _ = make_oracle(dtype=torch.float64, whiten=True, W=24, H=15, V=10, K=3, n_points=(5, 4), jitter=1e-6, lengthscale=0.25)
    for k in m2.params:
        m2.params[k].data.copy_(m.params[k].data)
    assert abs(m2.loss_and_grads(eps)[0] - loss) > 1e-3 * abs(loss)


def test_three_dimensional_world_with_learnable_inducing_points():
    """D = 3 (e.g. two spatial coordinates and time; the reference takes any len(world)): loss and every gradient, incl. the
    inducing inputs', fp64 against autograd."""
    from oracle.gdrf_oracle import RefShapedGDRF
    g = torch.Generator().manual_seed(21)
    N, V, K = 700, 9, 3
    xs = torch.rand(N, 3, generator=g, dtype=torch.float64)
    ws = torch.randint(0, 6, (N, V), generator=g, dtype=torch.int32)
    Z = 0.1 + 0.8 * torch.rand(18, 3, generator=g, dtype=torch.float64)
    m = RefShapedGDRF(xs, ws, kind="matern52", K=K, n_points=(3, 3, 2), Z=Z, lengthscale=0.4, jitter=1e-6, learn_inducing=True,
                      dtype=torch.float64)
    with torch.no_grad():
        m.params["u_loc"].add_(0.3 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64))
        m.params["phi_unc"].add_(0.5 * torch.randn(m.params["phi_unc"].shape, generator=g, dtype=torch.float64))
    eps = torch.randn(K, N, generator=g, dtype=torch.float64)
    loss, grads = m.loss_and_grads(eps)
    eng = engine_from_oracle(m)
    assert eng.D == 3
    eng.loss_and_grads(dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=m.last_jitter_level)
    assert abs(eng.read_out()["loss"] - loss) < LOSS_TOL_VS_TORCH * abs(loss)
    for name in eng.param_names:
        got, ref = eng.view(name, eng.grads).cpu().numpy(), grads[name].numpy()
        assert np.abs(got - ref).max() < 1e-8 * max(np.abs(ref).max(), 1e-6), name


# ---- odd shapes through the default fp32 build (bf16x6 kernels, two-group forms, DMA staging) -------------------------------
ODD_SHAPES = [
    dict(W=13, H=10, V=7, K=1, n_points=(3, 3)),           # K = 1: bwd_wbar's single-group fallback; N = 130: a padded second row tile
    dict(W=3, H=2, V=5, K=3, n_points=(4, 2)),             # six observations: far less than one row tile
    dict(W=43, H=3, V=9, K=2, n_points=(7, 5)),            # M = 35 -> Mp = 64, N = 129
    dict(W=30, H=20, V=11, K=17, n_points=(13, 10)),       # K = 17, M = 130 -> Mp = 160 (two column tiles, the second partial)
    dict(W=40, H=16, V=6, K=5, n_points=(17, 16), kind="matern52", lengthscale=0.1),   # M = 272 -> Mp = 288 (3 column tiles)
    dict(W=25, H=20, V=8, K=4, n_points=(12,), one_d=True, lengthscale=0.2),          # 1-D
]


@pytest.mark.parametrize("case", ODD_SHAPES, ids=lambda c: "x".join(str(v) for v in (c["W"] * c["H"], c["K"]) + tuple(c["n_points"])))
def test_default_fp32_build_on_odd_shapes(case):
    """Loss and gradients of the default build (fp32 arrays, f64 solve, exact-split bf16 contractions) against the fp64 oracle
    at identical parameters (both split modes), on shapes that leave partial tiles, padded tile pairs and the fallback kernels."""
    # a lengthscale of about one inducing spacing keeps K_uu + jitter well inside fp64: at 0.3 on a 13 x 10 grid the fp64
    # oracle itself returns q = k^T K_uu^-1 k = 25 > variance, and every gradient downstream is conditioning noise
    kw = dict(jitter=1e-4, lengthscale=0.08)
    kw.update(case)
    m, eps = make_oracle(dtype=torch.float64, **kw)
    loss, grads = m.loss_and_grads(eps)
    gmax = max(float(g.abs().max()) for g in grads.values())
    errs = {}
    for mode in ("f32",) + SPLIT_MODES:
        eng = engine_from_oracle(m, dtype=torch.float32, mfma_mode=mode, store_t=False)
        # same cumulative jitter on both sides: left alone, the engine decides the level in fp32 as the reference would
        eng.loss_and_grads(dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=m.last_jitter_level)
        out = eng.read_out()
        assert abs(out["loss"] - loss) < 2e-5 * abs(loss), mode
        errs[mode] = {}
        for name in eng.PARAM_NAMES:
            got, ref = eng.view(name, eng.grads).cpu().double().numpy(), grads[name].numpy()
            assert np.isfinite(got).all(), (mode, name)
            errs[mode][name] = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3 * gmax)
    # fp32 conditioning (var = s^2 - |w|^2 + tt cancels) bounds both builds alike; the emulated contractions must not add to it
    for mode in SPLIT_MODES:
        for name, e in errs[mode].items():
            assert e < max(3 * errs["f32"][name], 5e-3), (mode, name, errs)
            assert e < 5e-2, (mode, name, errs)


# ---- mean_function (abstract_gdrf.py:33-48; sparse_gdrf.py:346,395) -----------------------------------------------------
def _mean_per_topic(xs):          # (K, N): a different trend for every topic
    return torch.stack([(k + 1) * 0.7 * torch.sin(3.0 * xs[:, 0] + k) - 0.4 * k * xs[:, -1] for k in range(4)])


MEANS = {
    "per_topic_KxN": _mean_per_topic,
    "per_topic_Kx1": lambda xs: torch.arange(4, dtype=xs.dtype, device=xs.device).unsqueeze(-1) * 6.0 - 9.0,
    "reference_shape_N": lambda xs: 2.0 * xs[:, 0] - 1.0,      # xs.shape[:-1], like zero_mean: the softmax link cancels it
}


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("which", list(MEANS))
def test_mean_function_values_enter_the_link(which, dtype):
    fn = MEANS[which]
    m, eps = make_oracle(dtype=torch.float64, K=4, W=21, H=13, n_points=(5, 4), mean_function=fn)
    loss, grads = m.loss_and_grads(eps)
    m0, _ = make_oracle(dtype=torch.float64, K=4, W=21, H=13, n_points=(5, 4))
    loss0, _ = m0.loss_and_grads(eps)
    if which == "reference_shape_N":
        assert abs(loss - loss0) < 1e-12 * abs(loss0)          # a shift common to the K topics does not reach softmax(mu)
    else:
        assert abs(loss - loss0) > 2e-4 * abs(loss0)           # the case is not vacuous
    eng = engine_from_oracle(m, dtype=dtype)
    xs = dev(m.xs, eng)
    eng.loss_and_grads(xs, dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=m.last_jitter_level, mean=fn(xs))
    tol = 1e-9 if dtype == torch.float64 else 5e-3           # fp32: the hyper-parameter gradients carry the fp32 conditioning of the solve
    ltol = LOSS_TOL_VS_TORCH if dtype == torch.float64 else 2e-5
    assert abs(eng.read_out()["loss"] - loss) < ltol * abs(loss)
    gmax = max(float(g.abs().max()) for g in grads.values())
    for name in eng.PARAM_NAMES:
        got, ref = eng.view(name, eng.grads).cpu().double().numpy(), grads[name].numpy()
        assert np.abs(got - ref).max() < tol * max(np.abs(ref).max(), 1e-3 * gmax), name
    # the next call without a mean is the zero_mean step again (the borrowed pointer is dropped)
    eng.loss_and_grads(xs, dev(m.ws, eng, torch.int32), dev(eps, eng), force_level=m.last_jitter_level)
    assert abs(eng.read_out()["loss"] - loss0) < ltol * abs(loss0)
    with pytest.raises(ValueError):
        eng.loss_and_grads(xs, dev(m.ws, eng, torch.int32), dev(eps, eng), mean=torch.zeros(3, 7))

"""GPU: fidelity of the fp32 HIP path at the HEADLINE workload's conditioning (M = 512 inducing points on a
32 x 16 grid, lengthscale 0.1: K_uu is severely ill-conditioned and the jitter schedule escalates), on a
reduced N the fp64 oracle finishes in seconds.  Compared with the fp64 oracle evaluated at the same fp32
parameters, the same eps and the same jitter level (north_star tolerance: 1e-4 relative on the ELBO and on
posterior topic proportions).

Both quantities are compared AT IDENTICAL PARAMETERS.  Multi-step fp32 trajectories are not comparable between
any two fp32 implementations: Adam divides each gradient by its own magnitude, so components whose exact gradient
is ~0 turn rounding noise into +-lr updates (tests/golden/make_golden.py has the same remark); trajectory parity
is asserted in fp64 (tests/test_gpu_golden.py, test_gpu_parity.py::test_five_optimizer_steps_fp64)."""
import numpy as np
import pytest
import torch

from tests._util import dev, engine_from_oracle, make_oracle

pytestmark = pytest.mark.gpu


def _run(dtype, W, H, steps):
    m32, _ = make_oracle(kind="rbf", W=W, H=H, V=50, K=10, n_points=(32, 16), dtype=torch.float32, jitter=1e-6, perturb=True,
                         lr=1e-3, lengthscale=0.1)
    eng = engine_from_oracle(m32, dtype=dtype)
    xs, ws = dev(m32.xs, eng), dev(m32.ws, eng, torch.int32)
    # the reference-shaped oracle in fp64 on the SAME initial parameters (those are derived from the factorisation
    # at the level the fp32 path needs, so force that level on the fp64 side too)
    lvl = eng.factorize()
    m64, _ = make_oracle(kind="rbf", W=W, H=H, V=50, K=10, n_points=(32, 16), dtype=torch.float64, jitter=1e-6, perturb=True,
                         lr=1e-3, lengthscale=0.1, force_jitter_level=lvl)
    for k in m64.params:
        with torch.no_grad():
            m64.params[k].copy_(m32.params[k].detach().double())
    m64.Z = m32.Z.double()                  # the float32-rounded inducing grid, as the engine holds it
    for name in eng.PARAM_NAMES:
        eng.view(name).copy_(m64.params[name].detach().to(eng.dtype))
    g = torch.Generator().manual_seed(3)
    rel = []
    tp = eng.predict(xs, 1).cpu().double().numpy()            # at identical parameters, before any optimizer step
    tp_ref = m64.topic_probs().numpy()
    tp_abs = float(np.abs(tp - tp_ref).max())
    for s in range(steps):
        eps = torch.randn(10, m64.N, generator=g, dtype=torch.float64)
        loss_ref = m64.step(eps)
        eng.loss_and_grads(xs, ws, dev(eps, eng), force_level=lvl)
        eng.adam("adam", 1e-3)
        out = eng.read_out()
        rel.append(abs(out["loss"] - loss_ref) / abs(loss_ref))
        if dtype == torch.float32:                              # keep comparing at identical parameters
            for name in eng.PARAM_NAMES:
                eng.view(name).copy_(m64.params[name].detach().to(eng.dtype))
    return lvl, rel, tp_abs, tp_abs / float(tp_ref.max())


def test_fp32_elbo_and_topic_probs_within_1e4_of_fp64_oracle_at_headline_conditioning():
    lvl, rel, tp_abs, tp_rel = _run(torch.float32, 60, 50, 5)
    print("jitter level", lvl, "loss rel err per step", ["%.2e" % r for r in rel], "topic_probs abs/rel", tp_abs, tp_rel)
    assert lvl >= 1                                   # fp32 Cholesky needs the escalated jitter here
    assert max(rel) < 1e-4
    assert tp_rel < 1e-4


def test_pure_fp32_mode_shows_why_the_solve_runs_in_fp64():
    """GDRF_F32_PURE (all-fp32, the reference's literal arithmetic): the fp32 triangular solve loses ~2 digits of
    the topic proportions at this conditioning.  Documented, not asserted tight."""
    m32, eps = make_oracle(kind="rbf", W=60, H=50, V=50, K=10, n_points=(32, 16), dtype=torch.float32, jitter=1e-6, perturb=True,
                           lengthscale=0.1)
    pure = engine_from_oracle(m32, pure_fp32=True)
    mixed = engine_from_oracle(m32)
    xs = dev(m32.xs, pure)
    lvl = mixed.factorize()
    pure.factorize(force_level=lvl)
    m64, _ = make_oracle(kind="rbf", W=60, H=50, V=50, K=10, n_points=(32, 16), dtype=torch.float64, jitter=1e-6, perturb=True,
                         lengthscale=0.1, force_jitter_level=lvl)
    for k in m64.params:
        with torch.no_grad():
            m64.params[k].copy_(m32.params[k].detach().double())
    ref = m64.topic_probs().numpy()
    e_pure = float(np.abs(pure.predict(xs, 1).cpu().double().numpy() - ref).max())
    e_mixed = float(np.abs(mixed.predict(xs, 1).cpu().double().numpy() - ref).max())
    print("topic_probs abs error: pure fp32", e_pure, "default fp32 (f64 solve)", e_mixed)
    assert e_mixed < 1e-4 and e_mixed < e_pure


def test_fp64_path_matches_fp64_oracle_at_headline_conditioning():
    lvl, rel, tp_abs, tp_rel = _run(torch.float64, 40, 25, 3)
    print("fp64 jitter level", lvl, rel, tp_abs)
    assert max(rel) < 1e-6 and tp_rel < 1e-7

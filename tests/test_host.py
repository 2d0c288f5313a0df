"""CPU tests of the host layer: the C-ABI library loads and exports every declared symbol, and the
Python mirror of the reference surface validates its arguments like the reference does."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_of_the_header(hip_lib):
    from gdrf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gdrf_hip.h")).read()
    declared = set(re.findall(r"\b(gdrf_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert hip_lib.gdrf_version() == 1


def test_no_product_module_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gdrf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f


def test_engine_fails_loudly_without_a_hip_device():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from gdrf_amd._lib import GdrfHipError
    from gdrf_amd.engine import Engine
    with pytest.raises(GdrfHipError, match="no CPU path"):
        Engine(16, 4, 2, 5, 2)


def test_kernel_registry_and_arguments():
    from gdrf_amd.kernels import KERNEL_DICT, RBF, Matern52
    k = KERNEL_DICT["rbf"](input_dim=2, lengthscale=torch.tensor(0.1), variance=torch.tensor(25.0))
    assert isinstance(k, RBF) and k.to("cuda:0") is k and abs(float(k.lengthscale) - 0.1) < 1e-7
    assert isinstance(KERNEL_DICT["matern52"](input_dim=1), Matern52)
    assert KERNEL_DICT["matern32"](input_dim=1).kernel_id == 2 and KERNEL_DICT["exponential"](input_dim=1).kernel_id == 3
    rq = KERNEL_DICT["rationalquadratic"](input_dim=1, scale_mixture=torch.tensor(2.0))
    assert rq.kernel_id == 4 and float(rq.scale_mixture) == 2.0 and float(KERNEL_DICT["rationalquadratic"](input_dim=1).scale_mixture) == 1.0
    with pytest.raises(ValueError):
        KERNEL_DICT["rationalquadratic"](input_dim=1, scale_mixture=torch.tensor(0.0))
    with pytest.raises(ValueError):
        RBF(2, lengthscale=torch.tensor(-1.0))


def test_optimizer_registry_and_state_before_binding():
    from gdrf_amd.optim import OPTIMIZER_DICT, Adam, AdamW, ClippedAdam
    assert Adam({"lr": 0.01}).lr == 0.01 and AdamW({"lr": 0.01}).args["weight_decay"] == 1e-2
    assert ClippedAdam({"lr": 0.01, "lrd": 0.5}).args["clip_norm"] == 10.0
    with pytest.raises(NotImplementedError):
        OPTIMIZER_DICT["sgd"]({"lr": 0.1})
    with pytest.raises(ValueError):
        Adam({"lr": 0.1, "nonsense": 1})
    o = Adam({"lr": 0.1})
    o.set_state({"u_loc": {"step": 3}})
    assert o.get_state() == {"u_loc": {"step": 3}}


def test_objectives_and_scale_wrapper():
    from gdrf_amd import poutine
    from gdrf_amd.infer import OBJECTIVE_DICT, SVI, Trace_ELBO
    assert OBJECTIVE_DICT["graphelbo"](max_plate_nesting=1, vectorize_particles=True, num_particles=1).num_particles == 1
    assert OBJECTIVE_DICT["elbo"](num_particles=4).num_particles == 4
    with pytest.raises(ValueError):
        OBJECTIVE_DICT["elbo"](num_particles=0)
    r = OBJECTIVE_DICT["renyielbo"](max_plate_nesting=1, vectorize_particles=True, num_particles=3, alpha=2.0)
    assert r.alpha == 2.0 and r.num_particles == 3 and OBJECTIVE_DICT["renyielbo"]().num_particles == 2
    with pytest.raises(ValueError):
        OBJECTIVE_DICT["renyielbo"](alpha=1.0)
    sc = poutine.scale(scale=0.25)

    def f():
        return 7
    w = sc(f)
    assert w() == 7 and w.scale == 0.25
    with pytest.raises(ValueError):
        poutine.scale(scale=0.0)
    with pytest.raises(TypeError):
        SVI(model=w, guide=w, optim=None, loss=Trace_ELBO())


def test_model_rejects_options_outside_the_hot_path_before_touching_the_gpu():
    from gdrf_amd.kernels import RBF
    from gdrf_amd.models import GDRF_MODEL_DICT, SparseMultinomialGDRF
    k = RBF(2, lengthscale=torch.tensor(0.1), variance=torch.tensor(25.0))
    base = dict(num_observation_categories=5, num_topic_categories=2, world=[(0.0, 1.0)] * 2, kernel=k, dirichlet_param=0.01,
                n_points=[3, 3], fixed_inducing_points=True)
    with pytest.raises(TypeError):
        SparseMultinomialGDRF(**{**base, "link_function": "softmax"})          # a callable is accepted (gdrf_step_local_link); anything else is not
    with pytest.raises(TypeError):
        SparseMultinomialGDRF(**{**base, "mean_function": 1.0})
    with pytest.raises(TypeError):
        SparseMultinomialGDRF(**{**base, "randomize_metric": "best"})
    with pytest.raises(ValueError):
        SparseMultinomialGDRF(**{**base, "inducing_init": "hexagonal"})
    with pytest.raises(AssertionError):
        SparseMultinomialGDRF(**{**base, "dirichlet_param": -1.0})
    with pytest.raises(NotImplementedError):
        GDRF_MODEL_DICT["gdrf"]()


def test_normalise_index_matches_train_script():
    import numpy as np
    from gdrf_amd.data import normalise_index, synth_circles
    idx = np.array([[3, 10], [5, 30], [4, 20]])
    out = normalise_index(idx)
    assert out.min() == 0 and out.max() == 1 and np.allclose(out[2], [0.5, 0.5])
    xs, ws, topic = synth_circles(20, 10, 30, 5, seed=1)
    assert xs.shape == (200, 2) and ws.shape == (200, 30) and ws.dtype == np.int32
    assert ws.sum(1).min() >= 30 and ws.sum(1).max() < 300 and xs.min() == 0 and xs.max() == 1


@pytest.mark.parametrize("mode", ["uniform", "now", "exp", "uniform_now", "exp_now", "uniform_exp"])
def test_streaming_probabilities_follow_train_script(mode):
    """gdrf/train_script.py:406-450: each schedule, normalised; 'now' always picks the newest observation."""
    import numpy as np
    from gdrf_amd.train import streaming_probabilities
    for n in (1, 2, 7):
        p = streaming_probabilities(mode, n, streaming_weight=0.1, streaming_exp=1.0)
        assert p.shape == (n,) and abs(p.sum() - 1) < 1e-12 and (p >= 0).all()
        if mode == "now":
            assert p[-1] == 1.0
        if mode in ("exp", "exp_now", "uniform_now") and n > 1:
            assert p[-1] == p.max()
        if mode == "uniform":
            assert np.allclose(p, 1.0 / n)
    if mode == "uniform_now":
        p = streaming_probabilities(mode, 4, streaming_weight=0.2)
        assert np.allclose(p, [0.05, 0.05, 0.05, 0.85])
    with pytest.raises(ValueError):
        streaming_probabilities("sometimes", 3)

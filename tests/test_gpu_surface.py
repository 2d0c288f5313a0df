"""GPU: the reference's Python surface for the hot path, used the way gdrf/train_script.py uses it."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

from gdrf_amd.data import synth_circles

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(dtype=torch.float32, K=4, V=20, n_points=(6, 5), W=30, H=20, opt="adamw", kernel="rbf", seed=3, jitter=1e-6, device="cuda:0"):
    from gdrf_amd import poutine
    from gdrf_amd.infer import OBJECTIVE_DICT, SVI
    from gdrf_amd.kernels import KERNEL_DICT
    from gdrf_amd.models import GDRF_MODEL_DICT
    from gdrf_amd.optim import OPTIMIZER_DICT
    xs_np, ws_np, _ = synth_circles(W, H, V, K, seed=seed)
    xs = torch.from_numpy(xs_np).float().to(device)                 # train_script.py:264-268
    ws = torch.from_numpy(ws_np).int().to(device)
    world = list(zip(xs.min(dim=0).values.cpu().numpy().tolist(), xs.max(dim=0).values.cpu().numpy().tolist()))
    kern = KERNEL_DICT[kernel](input_dim=2, lengthscale=torch.tensor(0.2).to(device), variance=torch.tensor(25.0).to(device)).to(device)
    model = GDRF_MODEL_DICT["sparsemultinomialgdrf"](
        xs=xs, ws=ws, world=world, kernel=kern, num_observation_categories=V, device=device, num_topic_categories=K,
        dirichlet_param=0.01, n_points=list(n_points), fixed_inducing_points=True, inducing_init="grid", maxjitter=15,
        jitter=jitter, randomize_wt_matrix=False, dtype=dtype, seed=seed)
    optimizer = OPTIMIZER_DICT[opt]({"lr": 0.01})
    objective = OBJECTIVE_DICT["graphelbo"](max_plate_nesting=1, vectorize_particles=True, num_particles=1)
    scale = poutine.scale(scale=1.0 / len(xs))
    svi = SVI(model=scale(model.model), guide=scale(model.guide), optim=optimizer, loss=objective)
    return model, svi, optimizer, xs, ws


def test_known_answers_at_initialisation():
    model, svi, _, xs, ws = _build()
    tp = model.topic_probs(xs)
    assert tp.shape == (len(xs), model.K) and torch.allclose(tp, torch.full_like(tp, 1.0 / model.K), atol=1e-6)     # A.6(1)
    assert abs(float(model.perplexity(xs, ws).item()) - model.V) < 1e-3 * model.V                                  # A.6(2)
    assert torch.allclose(model.word_topic_matrix, torch.full((model.K, model.V), 1.0 / model.V, device=xs.device), atol=1e-6)
    assert model.word_probs(xs).shape == (len(xs), model.V) and model.log_topic_probs(xs).shape == (model.K, len(xs))
    assert abs(float(model.kernel_lengthscale) - 0.2) < 1e-6 and abs(float(model.kernel_variance) - 25.0) < 1e-4
    assert model.dims == 2
    ust = model.u_scale_tril
    assert torch.allclose(ust[0], ust[-1]) and float(ust[0].triu(1).abs().max()) == 0.0


def test_training_loop_like_train_script_improves_the_fit():
    model, svi, opt, xs, ws = _build()
    p0 = float(model.perplexity(xs, ws).item())
    losses = []
    for epoch in range(40):
        model.train()
        losses.append(svi.step(xs=xs, ws=ws, subsample=False))
        model.eval()
    p1 = float(model.perplexity(xs, ws).item())
    assert all(np.isfinite(losses)) and p1 < p0 and np.mean(losses[-5:]) < np.mean(losses[:5])
    # checkpoint surface: deepcopy(model).half(), state_dict round trip, optimizer state
    ckpt = {"model": copy.deepcopy(model).half(), "optimizer": opt.get_state()}
    sd = ckpt["model"].float().state_dict()
    assert set(sd) == {"_kernel.lengthscale_unconstrained", "_kernel.variance_unconstrained", "noise_unconstrained",
                       "u_loc_unconstrained", "_word_topic_matrix_map_unconstrained", "u_scale_tril_unconstrained"}
    model2, svi2, opt2, _, _ = _build()
    model2.load_state_dict(model.state_dict(), strict=False)
    opt2.set_state(opt.get_state())
    assert abs(float(model2.perplexity(xs, ws).item()) - p1) < 1e-5 * p1
    eps = torch.randn(model.K, len(xs), generator=torch.Generator().manual_seed(1))
    a = svi.step(xs=xs, ws=ws, subsample=False, eps=eps)
    b = svi2.step(xs=xs, ws=ws, subsample=False, eps=eps)
    assert abs(a - b) < 1e-6 * abs(a)
    assert opt2.get_state()["u_loc"]["step"] == opt.get_state()["u_loc"]["step"] == 41


def test_streaming_minibatch_steps_keep_the_global_scale():
    """train_script.py:394-465: tiny mini-batches, scale stays 1/len(full xs) (quirk Q9)."""
    model, svi, _, xs, ws = _build()
    rng = np.random.default_rng(0)
    for epoch in range(5):
        sel = rng.choice(epoch + 3, size=3)
        loss = svi.step(xs=xs[sel, ...], ws=ws[sel, ...], subsample=False)
        assert np.isfinite(loss)
    one = svi.step(xs=xs[5].unsqueeze(0), ws=ws[5].unsqueeze(0), subsample=False)          # streaming_size <= 1
    assert np.isfinite(one)


def test_inputs_outside_the_world_are_rejected():
    model, svi, _, xs, ws = _build()
    with pytest.raises(AssertionError):
        model.topic_probs(xs + 2.0)


def test_free_running_eps_is_reproducible_for_a_seed():
    m1, s1, _, xs, ws = _build(seed=9)
    m2, s2, _, _, _ = _build(seed=9)
    a = [s1.step(xs=xs, ws=ws) for _ in range(3)]
    b = [s2.step(xs=xs, ws=ws) for _ in range(3)]
    assert a == b


def _dist_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)          # both ranks share the box's single GPU
    model, svi, _, xs, ws = _build(dtype=torch.float64)
    N = len(xs)
    lo, hi = rank * N // world, (rank + 1) * N // world
    svi.row_offset = lo
    if os.environ.get("GDRF_TEST_C_ABI_ALLREDUCE") == "1":
        # the collective registered behind the C ABI (gdrf_set_allreduce / gdrf_payload_allreduce) instead of the engine's own
        # torch.distributed call: what a non-Python host would do with ncclAllReduce on its RCCL communicator
        eng = model._engine_for(hi - lo)
        eng.pg = None

        def allreduce(buf, count, is_double, stream):
            assert buf == eng.red_T.data_ptr() and count == eng.red_T.numel() and is_double == (eng.dtype == torch.float64)
            dist.all_reduce(eng.red_T)
            return 0
        eng.set_allreduce(allreduce)
    losses = [svi.step(xs=xs[lo:hi], ws=ws[lo:hi], subsample=False) for _ in range(3)]
    torch.save({"losses": losses, "params": model._engine.params.cpu()}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("via", ["torch_distributed", "c_abi_hook"])
def test_two_ranks_on_one_gpu_match_a_single_rank(tmp_path, via, monkeypatch):
    import torch.multiprocessing as mp
    monkeypatch.setenv("GDRF_TEST_C_ABI_ALLREDUCE", "1" if via == "c_abi_hook" else "0")
    port = 29600 + (os.getpid() % 2000) + (7 if via == "c_abi_hook" else 0)
    mp.spawn(_dist_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    model, svi, _, xs, ws = _build(dtype=torch.float64)
    ref = [svi.step(xs=xs, ws=ws, subsample=False) for _ in range(3)]
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert r0["losses"] == r1["losses"]
    assert np.allclose(r0["losses"], ref, rtol=1e-10)                      # Philox keyed by the global row: same eps
    assert torch.equal(r0["params"], r1["params"])
    assert (r0["params"] - model._engine.params.cpu()).abs().max() < 1e-9


def _rccl_worker(rank, world, port, tmp, dtype_name):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))   # RCCL, one GPU per rank
    dtype = getattr(torch, dtype_name)
    model, svi, _, xs, ws = _build(dtype=dtype, device=f"cuda:{rank}")
    N = len(xs)
    lo, hi = rank * N // world, (rank + 1) * N // world
    svi.row_offset = lo
    losses = [svi.step(xs=xs[lo:hi], ws=ws[lo:hi], subsample=False) for _ in range(3)]
    torch.save({"losses": losses, "params": model._engine.params.cpu()}, os.path.join(tmp, f"nccl_{dtype_name}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL all-reduce of the step (one rank per GPU)")
@pytest.mark.parametrize("dtype_name", ["float64", "float32"])
def test_two_ranks_over_rccl_match_a_single_rank(tmp_path, dtype_name):
    """The same sharded step as above with backend="nccl" (RCCL over xGMI), one rank per GPU: the step's single all-reduce of the
    flat payload (red_d packed into its tail).  Skipped on one-GPU boxes; the driver's multi-GPU node runs it."""
    import torch.multiprocessing as mp
    port = 32600 + (os.getpid() % 2000)
    mp.spawn(_rccl_worker, args=(2, port, str(tmp_path), dtype_name), nprocs=2, join=True)
    dtype = getattr(torch, dtype_name)
    model, svi, _, xs, ws = _build(dtype=dtype)
    ref = [svi.step(xs=xs, ws=ws, subsample=False) for _ in range(3)]
    r0 = torch.load(tmp_path / f"nccl_{dtype_name}_0.pt", weights_only=True)
    r1 = torch.load(tmp_path / f"nccl_{dtype_name}_1.pt", weights_only=True)
    assert r0["losses"] == r1["losses"] and torch.equal(r0["params"], r1["params"])      # replicated epilogue: bit-identical ranks
    tol = 1e-10 if dtype == torch.float64 else 1e-5
    assert np.allclose(r0["losses"], ref, rtol=tol)
    assert (r0["params"] - model._engine.params.cpu()).abs().max() < (1e-9 if dtype == torch.float64 else 1e-4)


def _build_lz_renyi(dtype=torch.float64):
    """Learnable inducing inputs + RenyiELBO(alpha=0.5, 2 particles): the payload entries only those options add."""
    from gdrf_amd import poutine
    from gdrf_amd.infer import SVI, OBJECTIVE_DICT
    from gdrf_amd.kernels import KERNEL_DICT
    from gdrf_amd.models import GDRF_MODEL_DICT
    from gdrf_amd.optim import OPTIMIZER_DICT
    xs_np, ws_np, _ = synth_circles(30, 20, 12, 3, seed=5)
    device = "cuda:0"
    xs, ws = torch.from_numpy(xs_np).to(device=device, dtype=dtype), torch.from_numpy(ws_np).int().to(device)
    kern = KERNEL_DICT["rbf"](input_dim=2, lengthscale=torch.tensor(0.2), variance=torch.tensor(25.0)).to(device)
    model = GDRF_MODEL_DICT["sparsemultinomialgdrf"](
        xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=kern, num_observation_categories=12, device=device, num_topic_categories=3,
        dirichlet_param=0.01, n_points=[5, 4], fixed_inducing_points=False, inducing_init="random", maxjitter=15, jitter=1e-6,
        dtype=dtype, seed=5)
    svi = SVI(model=poutine.scale(scale=1.0 / len(xs))(model.model), guide=poutine.scale(scale=1.0 / len(xs))(model.guide),
              optim=OPTIMIZER_DICT["adam"]({"lr": 0.02}),
              loss=OBJECTIVE_DICT["renyielbo"](max_plate_nesting=1, vectorize_particles=True, num_particles=2, alpha=0.5))
    return model, svi, xs, ws


def _dist_worker_lz(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, svi, xs, ws = _build_lz_renyi()
    N = len(xs)
    lo, hi = rank * N // world, (rank + 1) * N // world
    svi.row_offset = lo
    losses = [svi.step(xs=xs[lo:hi], ws=ws[lo:hi], subsample=False) for _ in range(3)]
    torch.save({"losses": losses, "params": model._engine.params.cpu()}, os.path.join(tmp, f"lz{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_with_learnable_inducing_points_and_renyi_match_a_single_rank(tmp_path):
    """The all-reduced payload also carries the inducing-input sums, and RenyiELBO's particle weights need the particles' ELBO
    over ALL ranks: two ranks (gloo, one GPU) must reproduce the single-rank losses and parameters."""
    import torch.multiprocessing as mp
    port = 31600 + (os.getpid() % 2000)
    mp.spawn(_dist_worker_lz, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    model, svi, xs, ws = _build_lz_renyi()
    ref = [svi.step(xs=xs, ws=ws, subsample=False) for _ in range(3)]
    r0 = torch.load(tmp_path / "lz0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "lz1.pt", weights_only=True)
    assert r0["losses"] == r1["losses"] and torch.equal(r0["params"], r1["params"])
    assert np.allclose(r0["losses"], ref, rtol=1e-9)
    assert (r0["params"] - model._engine.params.cpu()).abs().max() < 1e-8
    zoff = model._engine.layout["inducing_unc"]
    assert (r0["params"][zoff:zoff + 40] - model._engine.params.cpu()[zoff:zoff + 40]).abs().max() < 1e-8


def test_train_driver_full_batch_and_streaming():
    from gdrf_amd.train import train
    xs, ws, _ = synth_circles(24, 16, 20, 4, seed=2)
    out = train(xs=xs, ws=ws, dimensions=2, epochs=25, num_topics=4, num_inducing_points=[6, 4], inducing_initialization_method="grid",
                kernel_lengthscale=0.2, optimizer_type="adamw", optimizer_lr=0.01, jitter=1e-6)
    h = out["history"]
    assert h.shape == (25, 4) and np.isfinite(h).all() and h[-1, 1] < h[0, 1]          # perplexity improves
    xs1, ws1, _ = synth_circles(60, 1, 12, 3, seed=4, one_d=True)
    for mode in ("uniform_now", "exp"):
        out = train(xs=xs1, ws=ws1, dimensions=1, num_topics=3, num_inducing_points=8, inducing_initialization_method="grid",
                    streaming_inference=mode, streaming_size=3, kernel_lengthscale=0.2, optimizer_lr=0.01, jitter=1e-6,
                    perplexity_every=20)
        assert out["history"].shape[0] == 60 and np.isfinite(out["history"][:, 0]).all()


def test_learnable_inducing_points_like_the_reference_default():
    """SparseMultinomialGDRF(..., fixed_inducing_points=False) (the reference constructor's default, sparse_gdrf.py:24,79-88):
    the inducing inputs are an interval(0,1)-constrained parameter trained with everything else; random interior init."""
    from gdrf_amd import poutine
    from gdrf_amd.infer import SVI, OBJECTIVE_DICT
    from gdrf_amd.kernels import KERNEL_DICT
    from gdrf_amd.models import GDRF_MODEL_DICT
    from gdrf_amd.optim import OPTIMIZER_DICT
    xs_np, ws_np, _ = synth_circles(30, 20, 12, 3, seed=5)
    device = "cuda:0"
    xs, ws = torch.from_numpy(xs_np).float().to(device), torch.from_numpy(ws_np).int().to(device)
    kern = KERNEL_DICT["rbf"](input_dim=2, lengthscale=torch.tensor(0.2), variance=torch.tensor(25.0)).to(device)
    model = GDRF_MODEL_DICT["sparsemultinomialgdrf"](
        xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=kern, num_observation_categories=12, device=device,
        num_topic_categories=3, dirichlet_param=0.01, n_points=[5, 4], fixed_inducing_points=False, inducing_init="random",
        maxjitter=15, jitter=1e-6, seed=5)
    z0 = model.inducing_points.clone()
    assert z0.shape == (20, 2) and float(z0.min()) >= 0.0 and float(z0.max()) <= 1.0
    assert "_inducing_points_unconstrained" in model.state_dict() and len(model.parameters()) == 7
    svi = SVI(model=poutine.scale(scale=1.0 / len(xs))(model.model), guide=poutine.scale(scale=1.0 / len(xs))(model.guide),
              optim=OPTIMIZER_DICT["adam"]({"lr": 0.02}), loss=OBJECTIVE_DICT["elbo"](max_plate_nesting=1, num_particles=1))
    p0 = float(model.perplexity(xs, ws).item())
    losses = [svi.step(xs=xs, ws=ws, subsample=False) for _ in range(30)]
    z1 = model.inducing_points
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])
    assert float((z1 - z0).abs().max()) > 1e-4 and float(z1.min()) >= 0.0 and float(z1.max()) <= 1.0
    # (the mean-only perplexity moves by +-1e-3 around V in these 30 noisy steps - v = O(variance) as the Normal's scale, quirk Q1 -,
    # in either direction depending on rounding: the ELBO above is what must improve)
    p1 = float(model.perplexity(xs, ws).item())
    assert np.isfinite(p1) and abs(p1 - p0) < 0.02 * p0
    # round trip through the checkpoint surface
    sd = model.state_dict()
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    assert torch.equal(model.inducing_points, z1)


def test_mean_function_and_randomize_metric_through_the_model_surface():
    """The constructor's callables (sparse_gdrf.py:26,34-37): mean_function is evaluated on the scaled inputs every step,
    randomize_metric picks the initial word-topic matrix as abstract_gdrf.py:70-78 does (the last candidate beating the
    score of the Dirichlet-parameter matrix)."""
    from gdrf_amd import poutine
    from gdrf_amd.infer import OBJECTIVE_DICT, SVI
    from gdrf_amd.kernels import KERNEL_DICT
    from gdrf_amd.models import GDRF_MODEL_DICT
    from gdrf_amd.optim import OPTIMIZER_DICT
    K, V = 3, 12
    xs_np, ws_np, _ = synth_circles(20, 15, V, K, seed=5)
    xs, ws = torch.from_numpy(xs_np).float().cuda(), torch.from_numpy(ws_np).int().cuda()
    seen, scores, cands = [], [], []

    def mean(x):
        seen.append((tuple(x.shape), float(x.min()), float(x.max())))
        return torch.stack([2.0 * x[:, 0], -2.0 * x[:, 0], x[:, 1]])

    def metric(wt, model):
        assert tuple(wt.shape) == (K, V) and abs(float(wt.sum(-2).mean()) - 1.0) < 1e-5 and model.K == K
        scores.append(float(wt[0, 0]))                   # favour mass of taxon 0 on topic 0
        cands.append(wt.detach().clone())
        return scores[-1]

    def build(**extra):
        kern = KERNEL_DICT["rbf"](input_dim=2, lengthscale=torch.tensor(0.2), variance=torch.tensor(25.0))
        return GDRF_MODEL_DICT["sparsemultinomialgdrf"](
            xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=kern, num_observation_categories=V, device="cuda:0",
            num_topic_categories=K, dirichlet_param=0.01, n_points=[5, 4], fixed_inducing_points=True, inducing_init="grid",
            maxjitter=15, jitter=1e-6, seed=11, **extra)

    model = build(mean_function=mean, randomize_wt_matrix=True, randomize_metric=metric, randomize_iters=7)
    assert len(scores) == 8                                            # the Dirichlet-parameter matrix, then 7 candidates
    beating = [c for sc, c in zip(scores[1:], cands[1:]) if sc > scores[0]]
    assert beating, "seed gives no candidate above the uniform matrix"
    # stored through the stacked-simplex transform (abstract_gdrf.py:79-84): the candidate's rows renormalised over V
    chosen = beating[-1] / beating[-1].sum(-1, keepdim=True)
    assert torch.allclose(model.word_topic_matrix, chosen, atol=1e-6)
    plain = build()
    start = model.state_dict()
    scale = poutine.scale(scale=1.0 / len(xs))
    losses = {}
    for name, mdl in (("mean", model), ("plain", plain)):
        mdl.load_state_dict(start)                                     # same parameters (randomised word-topic matrix): only the mean differs
        svi = SVI(model=scale(mdl.model), guide=scale(mdl.guide), optim=OPTIMIZER_DICT["adam"]({"lr": 0.01}),
                  loss=OBJECTIVE_DICT["elbo"](max_plate_nesting=1, vectorize_particles=True, num_particles=1))
        eps = torch.randn(K, len(xs), generator=torch.Generator().manual_seed(3))
        losses[name] = svi.evaluate_loss(xs=xs, ws=ws, eps=eps)
        first = svi.step(xs=xs, ws=ws, eps=eps)
        assert abs(first - losses[name]) < 1e-6 * abs(first)
    assert seen and seen[0][0] == (len(xs), 2) and seen[0][1] >= 0.0 and seen[0][2] <= 1.0
    assert abs(losses["mean"] - losses["plain"]) > 1e-4 * abs(losses["plain"])
    # predictions ignore the mean, as the reference's log_topic_probs does (sparse_gdrf.py:161-186)
    n_calls = len(seen)
    model.topic_probs(xs)
    assert len(seen) == n_calls

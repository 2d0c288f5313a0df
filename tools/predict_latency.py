import sys, time, torch
sys.path.insert(0, "/root/repo")
from gdrf_amd.data import synth_circles
from gdrf_amd.kernels import RBF
from gdrf_amd.models import SparseMultinomialGDRF
xs_np, ws_np, _ = synth_circles(1000, 1000, 50, 10, seed=777)
xs = torch.from_numpy(xs_np).to("cuda:0", torch.float32).contiguous(); ws = torch.from_numpy(ws_np).to("cuda:0").contiguous()
m = SparseMultinomialGDRF(xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=RBF(input_dim=2, lengthscale=torch.tensor(0.1), variance=torch.tensor(25.0)),
                          num_observation_categories=50, num_topic_categories=10, dirichlet_param=0.01, n_points=[32, 16], fixed_inducing_points=True,
                          inducing_init="grid", maxjitter=15, jitter=1e-6, device="cuda:0", dtype=torch.float32, seed=777)
for f in (lambda: m.perplexity(xs, ws), lambda: m.topic_probs(xs)):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): r = f()
    torch.cuda.synchronize(); print("%.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), float(r.sum()) if r.numel() > 1 else float(r))

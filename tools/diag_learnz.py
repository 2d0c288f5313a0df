"""Diagnostic (GPU): perplexity trajectory of the learnable-inducing-points surface test (run with GDRF_ROWS_MFMA=0/1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gdrf_amd import poutine
from gdrf_amd.data import synth_circles
from gdrf_amd.infer import SVI, OBJECTIVE_DICT
from gdrf_amd.kernels import KERNEL_DICT
from gdrf_amd.models import GDRF_MODEL_DICT
from gdrf_amd.optim import OPTIMIZER_DICT
xs_np, ws_np, _ = synth_circles(30, 20, 12, 3, seed=5)
device = "cuda:0"
xs, ws = torch.from_numpy(xs_np).float().to(device), torch.from_numpy(ws_np).int().to(device)
for fixed in (False, True):
    kern = KERNEL_DICT["rbf"](input_dim=2, lengthscale=torch.tensor(0.2), variance=torch.tensor(25.0)).to(device)
    model = GDRF_MODEL_DICT["sparsemultinomialgdrf"](
        xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=kern, num_observation_categories=12, device=device,
        num_topic_categories=3, dirichlet_param=0.01, n_points=[5, 4], fixed_inducing_points=fixed, inducing_init="random",
        maxjitter=15, jitter=1e-6, seed=5)
    svi = SVI(model=poutine.scale(scale=1.0 / len(xs))(model.model), guide=poutine.scale(scale=1.0 / len(xs))(model.guide),
              optim=OPTIMIZER_DICT["adam"]({"lr": 0.02}), loss=OBJECTIVE_DICT["elbo"](max_plate_nesting=1, num_particles=1))
    pp = [float(model.perplexity(xs, ws).item())]
    ls = []
    for i in range(120):
        ls.append(svi.step(xs=xs, ws=ws, subsample=False))
        if i % 10 == 9: pp.append(float(model.perplexity(xs, ws).item()))
    print("fixed" if fixed else "learn", os.environ.get("GDRF_ROWS_MFMA", "1"), "perplexity:", ["%.5f" % p for p in pp], "loss first/last 5: %.4f %.4f" % (np.mean(ls[:5]), np.mean(ls[-5:])))

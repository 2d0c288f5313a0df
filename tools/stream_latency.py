"""Latency of one streaming mini-batch step (train_script.py:394-465: tiny-N svi.step calls) on the engine: wall time per
step with the host reading the loss every step, as the reference's loop does, plus the in-library kernel timers."""
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gdrf_amd.engine import Engine
M, K, V, D = 512, 10, 50, 2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine(n, M, K, V, D, dtype=torch.float32, kernel="rbf", device="cuda:0", jitter=1e-6, maxjitter=15)
g = torch.Generator().manual_seed(0)
ax = torch.linspace(0, 1, 32, dtype=torch.float64); ay = torch.linspace(0, 1, 16, dtype=torch.float64)
Z = torch.stack([t.flatten() for t in torch.meshgrid(ax, ay, indexing="ij")]).T
eng.set_inducing_points(Z); eng.set_dirichlet(torch.full((K, V), 0.01, dtype=torch.float64))
eng.view("log_lengthscale").fill_(float(np.log(0.1))); eng.view("log_variance").fill_(float(np.log(25.0))); eng.view("log_noise").fill_(0.0)
eng.view("phi_unc").copy_(torch.randn(K, V, generator=g))
eng.factorize()
L = eng.workspace("L").to(eng.dtype)
eng.view("u_scale_tril_unc").copy_((L.tril(-1) + torch.diag(L.diagonal().log())).unsqueeze(0).expand(K, -1, -1))
xs = torch.rand(n, D, generator=g).cuda(); ws = torch.randint(0, 20, (n, V), generator=g, dtype=torch.int32).cuda()
eps = torch.randn(K, n, generator=g).cuda()
def step():
    eng.loss_and_grads(xs, ws, eps, n_global=1e6)
    eng.adam("adam", 1e-5)
    return eng.read_out()["loss"]
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 50
for _ in range(R): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
print("n=%d: %.3f ms per step (host reads the loss each step)" % (n, dt * 1e3))
# host enqueue time alone (no read of the loss inside the loop): what a captured graph would remove
torch.cuda.synchronize(); h = 0.0
for _ in range(R):
    t1 = time.perf_counter()
    eng.loss_and_grads(xs, ws, eps, n_global=1e6)
    eng.adam("adam", 1e-5)
    h += time.perf_counter() - t1
    torch.cuda.synchronize()
print("host enqueue per step: %.3f ms" % (h / R * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(R):
    eng.loss_and_grads(xs, ws, eps, n_global=1e6)
    eng.adam("adam", 1e-5)
torch.cuda.synchronize()
print("back-to-back without reading the loss: %.3f ms per step" % ((time.perf_counter() - t0) / R * 1e3))
eng.set_timing(True)
for _ in range(20): step()
torch.cuda.synchronize()
tm = eng.get_timing()
print("in-library timers (ms per step): " + " ".join("%s %.3f" % (k, v["ms"] / max(v["count"], 1)) for k, v in tm.items() if v["count"]))

"""Sweep launch variants of the standalone K_nm kernel (grid cap, nontemporal vs plain stores) on the GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdrf_amd.engine import Engine
N, M = 1_000_000, 512
eng = Engine(N, M, 2, 5, 2, dtype=torch.float32, process_group=None)
g = torch.Generator().manual_seed(0)
eng.set_inducing_points(torch.rand(M, 2, generator=g))
eng.view("log_lengthscale").fill_(-2.3); eng.view("log_variance").fill_(3.2)
xs = torch.rand(N, 2, generator=g).cuda()
out = torch.empty(N, M, device="cuda")
for plain in ("0",):
    for blocks in (16384, 32768):
        os.environ["GDRF_KNM_PLAIN_STORES"] = plain
        os.environ["GDRF_KNM_BLOCKS"] = str(blocks)
        for _ in range(3):
            eng.knm_into(xs, out)
        eng.set_timing(True)
        for _ in range(20):
            eng.knm_into(xs, out)
        torch.cuda.synchronize()
        t = eng.get_timing()["k_nm"]
        eng.set_timing(False)
        ms = t["ms"] / t["count"]
        print(f"plain={plain} blocks={blocks:6d}  {ms:.4f} ms  {N*M*4/ms/1e6:.0f} GB/s  frac={N*M*4/ms/1e6/8000:.3f}", flush=True)

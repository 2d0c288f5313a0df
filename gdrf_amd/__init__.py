"""gdrf_amd: MI355X-native (gfx950) implementation of the GDRF SVI ELBO hot path.

Mirrors the reference's Python surface for that path (san-soucie/gdrf: gdrf.models,
pyro.infer.SVI, pyro.optim, pyro.contrib.gp.kernels as used by gdrf/train_script.py) on top of
hand-written HIP kernels reached through a C ABI (include/gdrf_hip.h).
"""
__version__ = "0.1.0"

"""Stand-ins for the pyro.contrib.gp.kernels objects the reference builds in
gdrf/train_script.py:93-99,289-298 (``KERNEL_DICT[name](input_dim=, lengthscale=, variance=)``).

They only carry the hyper-parameters into the model; the covariance arithmetic is the HIP
kernels' (gdrf_amd/csrc/common.h::cov_from_r2; pyro semantics restated in SURVEY.md A.3).
"""
from __future__ import annotations

import torch


class Kernel:
    name = None
    kernel_id = None

    def __init__(self, input_dim: int, variance=None, lengthscale=None, active_dims=None):
        if active_dims is not None and list(active_dims) != list(range(input_dim)):
            raise NotImplementedError("active_dims other than all input dimensions")
        self.input_dim = int(input_dim)
        self.variance = torch.as_tensor(1.0 if variance is None else variance, dtype=torch.float64).detach().cpu().reshape(())
        self.lengthscale = torch.as_tensor(1.0 if lengthscale is None else lengthscale, dtype=torch.float64).detach().cpu().reshape(())
        if not (self.variance > 0 and self.lengthscale > 0):
            raise ValueError("variance and lengthscale must be positive")

    def to(self, device):
        return self

    def __repr__(self):
        return f"{type(self).__name__}(input_dim={self.input_dim}, lengthscale={float(self.lengthscale)}, variance={float(self.variance)})"


class RBF(Kernel):
    name = "rbf"
    kernel_id = 0


class Matern52(Kernel):
    name = "matern52"
    kernel_id = 1


class Matern32(Kernel):
    name = "matern32"
    kernel_id = 2


class Exponential(Kernel):
    name = "exponential"
    kernel_id = 3


class RationalQuadratic(Kernel):
    """variance * (1 + r2 / (2 scale_mixture))^(-scale_mixture); scale_mixture is a third positive, learnable
    hyper-parameter (pyro.contrib.gp.kernels.RationalQuadratic(input_dim, variance, lengthscale, scale_mixture))."""
    name = "rationalquadratic"
    kernel_id = 4

    def __init__(self, input_dim: int, variance=None, lengthscale=None, scale_mixture=None, active_dims=None):
        super().__init__(input_dim, variance=variance, lengthscale=lengthscale, active_dims=active_dims)
        self.scale_mixture = torch.as_tensor(1.0 if scale_mixture is None else scale_mixture,
                                             dtype=torch.float64).detach().cpu().reshape(())
        if not self.scale_mixture > 0:
            raise ValueError("scale_mixture must be positive")


KERNEL_DICT = {"rbf": RBF, "matern32": Matern32, "matern52": Matern52, "exponential": Exponential,
               "rationalquadratic": RationalQuadratic}

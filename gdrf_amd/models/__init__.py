"""Model surface of the reference's ``gdrf.models`` (gdrf/models/__init__.py:1-18) for the SVI hot
path.  Only ``SparseMultinomialGDRF`` (the north-star model) is built; the other exported names
exist so that registries written against the reference fail with a clear message."""
from .sparse_gdrf import ModelSnapshot, SparseMultinomialGDRF, validate_dirichlet_param


def _out_of_scope(name, why):
    class _X:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} is outside this build's hot path: {why}")
    _X.__name__ = name
    return _X


SparseGDRF = _out_of_scope("SparseGDRF", "Categorical-z variant, not the Multinomial path (SURVEY.md section 2 row 1)")
GDRF = _out_of_scope("GDRF", "dense N x N model, O(N^3) (SURVEY.md section 2 row 5)")
MultinomialGDRF = _out_of_scope("MultinomialGDRF", "dense N x N model, O(N^3) (SURVEY.md section 2 row 5)")
SimpleGDRF = _out_of_scope("SimpleGDRF", "block-diagonal KN x KN experimental variant (SURVEY.md section 2 row 6)")
SimpleMultinomialGDRF = _out_of_scope("SimpleMultinomialGDRF", "block-diagonal KN x KN experimental variant")

GDRF_MODEL_DICT = {
    "gdrf": GDRF, "multinomialgdrf": MultinomialGDRF, "sparsegdrf": SparseGDRF,
    "sparsemultinomialgdrf": SparseMultinomialGDRF, "simplegdrf": SimpleGDRF,
    "simplemultinomialgdrf": SimpleMultinomialGDRF,
}

__all__ = ["GDRF", "MultinomialGDRF", "SparseGDRF", "SparseMultinomialGDRF", "SimpleGDRF", "SimpleMultinomialGDRF",
           "GDRF_MODEL_DICT", "ModelSnapshot", "validate_dirichlet_param"]

"""Model descriptors: the host-side names of the static Unfold kernels compiled into the library.

In modppl a model is a `dyngen!` function wrapped in `DynUnfold` (modppl/src/modeling/dynunfold.rs:7-18);
closures cannot cross the C ABI, so a model is selected by {kind, dims, params}.
"""
import numpy as np

from . import capi


class UnfoldModel:
    """Counterpart of `DynUnfold<State>`: a time-unrolled kernel the particle filter can extend."""

    def __init__(self, kind, dim_state, dim_obs, params, name):
        self.kind, self.dim_state, self.dim_obs, self.name = kind, dim_state, dim_obs, name
        self.params = np.ascontiguousarray(params, dtype=np.float64)

    def desc(self):
        import ctypes as C

        return capi.ModelDesc(self.kind, self.dim_state, self.dim_obs, len(self.params),
                              self.params.ctypes.data_as(C.POINTER(C.c_double)))

    def __repr__(self):
        return f"UnfoldModel({self.name}, params={self.params.tolist()})"


def lgssm_model(mu0=0.0, sig0=1.0, a=0.9, sig_x=0.5, sig_y=1.0):
    """Linear-Gaussian SSM, d=1 (BASELINE.json configs 1-2):
    t==0: x ~ normal(mu0, sig0); t>0: x ~ normal(a*x_prev, sig_x); y ~ normal(x, sig_y) observed."""
    return UnfoldModel(capi.MP_MODEL_LGSSM1, 1, 1, [mu0, sig0, a, sig_x, sig_y], "lgssm1")

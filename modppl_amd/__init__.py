"""modppl_amd — the MI355X-native SMC / importance / MH hot path of agarret7/modppl.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/modppl_hip.h) and the host-side
mirror of the reference's inference entry points.
"""
from .capi import ModpplError  # noqa: F401
from .inference import (FunctionChains, HierarchicalChains, ParticleSystem, PointedChains, fn_importance_resampling, fn_importance_sampling,  # noqa: F401
                        importance_resampling, importance_sampling, simulate)
from .models import (UnfoldModel, bearings_model, hmm_model, lgssm_band_model, lgssm_dense_model, lgssm_model, line_model,  # noqa: F401
                     pointed_2d_model, spiral_model, stochastic_volatility_model)

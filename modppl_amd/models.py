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


def spiral_model():
    """`spiral_model` of modppl/tests/dyngenfns/unfold.rs:14-32: polar random walk observed through
    mvnormal(pos, 0.001 I).  State (r, theta); init_step's `args` is the (unused) initial state."""
    return UnfoldModel(capi.MP_MODEL_SPIRAL, 2, 2, [], "spiral")


def hmm_model(prior, emission_matrix, transition_matrix):
    """`hmm::HMM` of modppl/tests/hmm/model.rs.  Matrices column-stochastic as the reference builds them
    (`dmatrix![...].transpose()`): emission_matrix[o, s] = p(o | s), transition_matrix[s2, s1] = p(s2 | s1)."""
    prior = np.asarray(prior, dtype=np.float64)
    e = np.asarray(emission_matrix, dtype=np.float64)
    t = np.asarray(transition_matrix, dtype=np.float64)
    S, O = prior.size, e.shape[0]
    assert e.shape == (O, S) and t.shape == (S, S)
    return UnfoldModel(capi.MP_MODEL_HMM, 1, 1, np.concatenate([[S, O], prior, e.reshape(-1), t.reshape(-1)]), "hmm")


def bearings_model(p0x=1.0, p0y=1.0, sig_p0=1.0, sig_v0=0.1, sig_a=0.05, sig_theta=0.02):
    """Bearings-only tracker, d=4 (BASELINE.json config 3): constant velocity + accel noise, theta = atan2(py, px) + noise."""
    return UnfoldModel(capi.MP_MODEL_BEARINGS, 4, 1, [p0x, p0y, sig_p0, sig_v0, sig_a, sig_theta], "bearings")


def lgssm_band_model(D=16, a=0.9, band=0.05, sig0=1.0, sig_x=0.5, sig_y=1.0):
    """Banded LGSSM d=D (BASELINE.json config 5): x' = a (I + band B) x + sig_x z, y = x + sig_y e."""
    return UnfoldModel(capi.MP_MODEL_LGSSM_BAND, D, D, [D, a, band, sig0, sig_x, sig_y], f"lgssm_band{D}")


def pointed_2d_model(bounds=(-5.0, 5.0, -5.0, 5.0), cov=((1.0, -0.6), (-0.6, 2.0))):
    """`pointed_2d_model` of modppl/tests/dyngenfns/simple.rs:27-34 (static: one generate, n_steps = 1):
    latent ~ uniform_2d(bounds), obs ~ mvnormal(latent, cov) observed.  Defaults: tests/importance.rs:24-25."""
    c = np.asarray(cov, dtype=np.float64).reshape(4)
    return UnfoldModel(capi.MP_MODEL_POINTED_2D, 2, 2, list(bounds) + list(c), "pointed_2d")


def line_model(xs=tuple(float(v) for v in range(-5, 6))):
    """`line_model` of modppl/tests/dyngenfns/simple.rs:9-24 (static): slope ~ normal(0,1), intercept ~ normal(0,2),
    ys/i ~ normal(slope x_i + intercept, 0.1) observed.  Default design: tests/importance.rs:62."""
    return UnfoldModel(capi.MP_MODEL_LINE, 2, len(xs), list(xs), "line")


def lgssm_dense_model(A, Q, R, sig0=1.0):
    """Dense LGSSM, d = 16: t==0: x ~ mvnormal(0, sig0^2 I); t>0: x ~ mvnormal(A x_prev, Q); mvnormal(x, R) observed
    (two `mvnormal` sites, modppl/src/modeling/dists/mvnormal.rs:14-38; Q may be singular: the eigen `transform`, :30-33).
    The three 16 x 16 products per particle-step run on the matrix cores (v_mfma_f64_16x16x4_f64)."""
    A, Q, R = (np.ascontiguousarray(m, dtype=np.float64) for m in (A, Q, R))
    D = A.shape[0]
    assert A.shape == Q.shape == R.shape == (D, D)
    return UnfoldModel(capi.MP_MODEL_LGSSM_DENSE, D, D, np.concatenate([[D, sig0], A.reshape(-1), Q.reshape(-1), R.reshape(-1)]), f"lgssm_dense{D}")


MP_MODEL_STOCHVOL = 100   # registered through MP_REGISTER_UNFOLD_MODEL (modppl_amd/csrc/mp_models_extra.h)


def stochastic_volatility_model(mu=-1.0, phi=0.95, sigma=0.25, sig0=0.8):
    """Stochastic volatility: h_0 ~ normal(mu, sig0); h_t ~ normal(mu + phi (h_{t-1} - mu), sigma); y_t ~ normal(0, exp(h_t / 2))
    observed.  The example model of the registration layer: its whole definition is one block of mp_models_extra.h."""
    return UnfoldModel(MP_MODEL_STOCHVOL, 1, 1, [mu, phi, sigma, sig0], "stochvol")

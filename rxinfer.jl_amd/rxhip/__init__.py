"""rxhip — host-side mirror of the RxInfer `infer(...)` surface for the MI355X engine.

The arithmetic lives in csrc/ (hand-written HIP for gfx950, exported through the C ABI in
include/rxhip.h); this package only marshals arguments.  Importing it does not require a GPU;
constructing an engine does (no CPU fallback)."""
from ._lib import RxHipError, lib, LIB_PATH  # noqa: F401
from .engine import LGSSMEngine, LGSSMNoiseEngine, GMMEngine, MvGMMEngine, HGFEngine, DriftChainEngine, Communicator  # noqa: F401
from .api import (InferenceResult, infer, linear_gaussian_ssm, MvNormalMeanCovariance, NormalMeanVariance,  # noqa: F401
                  GammaShapeRate, GammaShapeScale, Dirichlet, Wishart, gaussian_mixture, multivariate_gaussian_mixture, iid_normal_gamma,
                  hierarchical_gaussian_filter, univariate_drift_chain, time_varying_gaussian_ssm)

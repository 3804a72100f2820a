"""MI355X-native batched Kalman filter / RTS smoother log-likelihood.

Drop-in for the numeric hot path of jessegrabowski/pymc_statespace
(``pymc_statespace.filters``): hand-written HIP kernels for gfx950 behind a plain C-ABI
(``include/kf.h`` -> ``libkf_hip.so``), driven from Python through ctypes.  Importing this
package loads the HIP library and FAILS if it has not been built -- there is no CPU path.
"""
from . import _lib

_lib.load()  # raises KfLibraryMissing when libkf_hip.so is absent: no silent fallback

from .engine import FilterResult, KalmanEngine, KalmanError, get_engine  # noqa: E402
from .filters import (  # noqa: E402
    FILTER_FACTORY,
    CholeskyFilter,
    KalmanSmoother,
    SingleTimeseriesFilter,
    StandardFilter,
    SteadyStateFilter,
    UnivariateFilter,
)

__all__ = [
    "KalmanEngine", "KalmanError", "FilterResult", "get_engine", "StandardFilter",
    "CholeskyFilter", "UnivariateFilter", "SingleTimeseriesFilter", "SteadyStateFilter", "KalmanSmoother",
    "FILTER_FACTORY",
]

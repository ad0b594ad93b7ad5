"""gpcsd_amd -- MI355X-native GPCSD hot path behind the reference's Python surface.

Module and class names follow natalieklein/gpcsd (`gpcsd1d.GPCSD1D`, `gpcsd2d.GPCSD2D`, `covariances.*`,
`forward_models.*`, `utility_functions.*`, `priors.*`) so existing scripts switch by changing the import.
All array arithmetic runs in libgpcsd_hip.so (hand-written HIP for gfx950) through ctypes; there is no CPU
fallback -- compute calls raise if the library or a GPU is missing.  Nothing touches the GPU at import time.
"""
from . import _hip  # noqa: F401
from . import priors, forward_models, utility_functions, covariances, predict_csd  # noqa: F401
from . import gpcsd1d, gpcsd2d, dist  # noqa: F401
from .gpcsd1d import GPCSD1D  # noqa: F401
from .gpcsd2d import GPCSD2D  # noqa: F401
from ._hip import GPCSDCapacityError, HipUnavailable  # noqa: F401

__version__ = "0.1.0"

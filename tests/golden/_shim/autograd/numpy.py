"""`autograd.numpy` -> NumPy (forward-only shim, see package docstring)."""
import sys
import numpy as _np

sys.modules[__name__] = _np

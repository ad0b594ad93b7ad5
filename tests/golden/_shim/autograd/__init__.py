"""Forward-only stand-in for the HIPS `autograd` package (not installed here, no network).

Ours, not reference code: it lets `tests/golden/generate_goldens.py` import the
reference package from /root/reference/src in THIS container so golden vectors
can be produced.  `autograd.numpy` is mapped to plain NumPy; `grad` raises,
because reverse mode is not available.
"""


def grad(fun, argnum=0):
    def _no_grad(*args, **kwargs):
        raise NotImplementedError("autograd shim is forward-only")
    return _no_grad

"""Identity-decorator stand-in for `numba`, used ONLY by tests/golden/make_golden.py.

The reference (jseppanen/azalea) decorates its Hex rules with numba's
``@jitclass`` / ``@njit``.  numba is not installable in the build container, but
the decorated bodies are valid pure Python/numpy, so importing the reference
with these no-op decorators runs the *unmodified* reference algorithm (slowly).
This file is test tooling written for this repo; it contains no reference code
and is never imported by the product (azalea_amd) or shipped to the GPU box's
run-time path.
"""
import numpy as _np


class _Type:
    """Callable + subscriptable wrapper over a numpy scalar type (e.g. int32[:, :])."""

    def __init__(self, np_type):
        self._t = np_type

    def __call__(self, x):
        return self._t(x)

    def __getitem__(self, _):
        return self

    def __repr__(self):
        return "shim.%s" % self._t.__name__


int32 = _Type(_np.int32)
uint32 = _Type(_np.uint32)
int64 = _Type(_np.int64)
float32 = _Type(_np.float32)
float64 = _Type(_np.float64)


def _identity_decorator_factory(*args, **kwargs):
    # used as @njit, @njit('sig'), @jit(nopython=True) ...
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn
    return deco


njit = _identity_decorator_factory
jit = _identity_decorator_factory


def jitclass(spec=None):
    def deco(cls):
        return cls
    return deco

"""oracle.twin -- the oracle with a FLOAT64 TWIN behind every Float32 call (test infrastructure, like everything under oracle/).

    from oracle import twin as O          # instead of: from oracle import oracle as O

Every name of oracle.oracle is available unchanged.  When a routine is called on Float32 data it is run a second time in Float64 on
THE SAME Float32 data (arrays and scalars widened exactly; index arrays, flags and Python floats as they are), state for state: the
arrays a routine returns or updates in place keep a Float64 twin that the next call continues from.  `twin_of(x)` hands out the
twin of an oracle result; tests/test_gpu_parity.close() uses it for the accuracy half of the Float32 tolerance statement (the
device's Float32 result against the Float64 value, at most 1e-4 = 840 eps32 -- the Float32 oracle's own sequential folds are
often the less accurate of the two).  Nothing here is used by the product."""
import sys

import numpy as np

from . import oracle as _O

_F32 = np.dtype(np.float32)
_C64 = np.dtype(np.complex64)
_twins = {}   # id(object) -> (object, its Float64 twin): the object is kept so that an id is never reused while it is registered


def _register(obj, twin):
    if len(_twins) > 20000:    # a test session creates a few thousand; old entries only pin small arrays
        _twins.clear()
    _twins[id(obj)] = (obj, twin)


def twin_of(x):
    """The Float64 twin of a Float32 oracle object (array, scalar, Problem, Prox, SepQuad), or None."""
    e = _twins.get(id(x))
    return e[1] if e is not None and e[0] is x else None


def _is32(x):
    return (isinstance(x, np.ndarray) and x.dtype in (_F32, _C64)) or isinstance(x, (np.float32, np.complex64))


def _up(x):
    """The Float64 view of one argument: its registered twin, or the exact widening of Float32 data seen for the first time."""
    t = twin_of(x)
    if t is not None:
        return t
    if isinstance(x, np.ndarray) and x.dtype in (_F32, _C64):
        t = x.astype(np.float64 if x.dtype == _F32 else np.complex128)
        _register(x, t)
        return t
    if isinstance(x, np.float32):
        return np.float64(x)
    if isinstance(x, np.complex64):
        return np.complex128(x)
    if isinstance(x, (list, tuple)) and any(_is32(e) for e in x):
        return type(x)(_up(e) for e in x)
    return x


def _has32(args, kwargs):
    for a in list(args) + list(kwargs.values()):
        if _is32(a) or twin_of(a) is not None or (isinstance(a, (list, tuple)) and any(_is32(e) for e in a)):
            return True
        if isinstance(a, (_O.Problem, _O.SepQuad)) and getattr(a, "dtype", None) == _F32:
            return True
    return False


def _pair(res, res64):
    if isinstance(res, tuple) and isinstance(res64, tuple):
        for r, r64 in zip(res, res64):
            _pair(r, r64)
    elif _is32(res) and res64 is not None:
        _register(res, res64)


def _wrap_function(fn):
    def call(*args, **kwargs):
        res = fn(*args, **kwargs)
        if _has32(args, kwargs):
            a64 = [_up(a) for a in args]
            k64 = {k: _up(v) for k, v in kwargs.items()}
            if k64.get("dtype") is np.float32:
                k64["dtype"] = np.float64
            try:
                res64 = fn(*a64, **k64)
            except Exception:   # (a routine that asserts on mixed types: no twin for this call)
                return res
            _pair(res, res64)
        return res
    call.__name__ = getattr(fn, "__name__", "call")
    call.__doc__ = fn.__doc__
    return call


def _wrap_class(cls):
    def make(*args, **kwargs):
        obj = cls(*args, **kwargs)
        is32 = _has32(args, kwargs) or kwargs.get("dtype") in (np.float32, _F32)
        if is32:
            a64 = [_up(a) for a in args]
            k64 = {k: _up(v) for k, v in kwargs.items()}
            if "dtype" in k64:
                k64["dtype"] = np.float64
            _register(obj, cls(*a64, **k64))
        return obj
    make.__name__ = cls.__name__
    return make


def assign(dst, src):
    """dst[:] = src, on the twin as well: a test that edits oracle state between two calls (a warm restart behind the library's
    back) must make the same edit on the Float64 twin, or the twin goes on from the unedited state."""
    dst[:] = src
    t = twin_of(dst)
    if t is not None:
        ts = twin_of(src)
        t[:] = ts if ts is not None else np.asarray(src, dtype=t.dtype)


def scale_inplace(dst, c):
    """dst *= c, on the twin as well (c: a Python float or a scalar of dst's type)."""
    dst *= dst.dtype.type(c)
    t = twin_of(dst)
    if t is not None:
        t *= t.dtype.type(dst.dtype.type(c))


_CLASSES = {"Problem", "Prox", "SepQuad"}
_PLAIN = {"build", "lib", "as_pairs", "as_complex", "julia_sum_vec", "julia_sum_scalar"}   # no state to twin


# Routines whose Float64 run is NOT a rounding-free version of the Float32 one but another run of the algorithm: adaptive Finito's
# backtracking test f_i(z) <= model + tol (Finito_adaptive.jl:125-131) sits on the boundary to within Float32 rounding often enough that
# a Float64 run takes other decisions than EVERY Float32 run, the reference's included -- a stepsize then differs by the factor 0.8
# and the trajectories part.  They run untwinned, and the state they touch loses its twin.
_UNTWINNED = {"afinito_steps"}


def _wrap_untwinned(fn):
    def call(*args, **kwargs):
        res = fn(*args, **kwargs)
        for a in list(args) + list(kwargs.values()):
            if _is32(a):
                _twins.pop(id(a), None)
        return res
    call.__name__ = getattr(fn, "__name__", "call")
    return call


def __getattr__(name):   # PEP 562: every other name of oracle.oracle, wrapped
    v = getattr(_O, name)
    if name in _UNTWINNED:
        w = _wrap_untwinned(v)
    elif name in _CLASSES:
        w = _wrap_class(v)
    elif callable(v) and not name.startswith("_") and name not in _PLAIN and not isinstance(v, type):
        w = _wrap_function(v)
    else:
        w = v
    setattr(sys.modules[__name__], name, w)
    return w
